OUT=gpurun_out/${1:-r4s}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/gpu_stamps_ukf.py > $OUT/ukf_phase_stamps.txt 2>&1
python3 tools/gpu_stamps.py > $OUT/ekf_phase_stamps.txt 2>&1
for W in ukf64 ekf64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$W -o s -- python3 bench.py --workload $W --no-legs --cpu-sample 0 > $OUT/bench_$W.json 2> $OUT/st_$W.err
find $OUT/st_$W -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$W.csv \;
rm -rf $OUT/st_$W
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p_${W}_${C} -o p -- python3 bench.py --workload $W --no-legs --cpu-sample 0 --steps 2 --warmup 1 > /dev/null 2> $OUT/p_${W}_${C}.err
  python3 - $OUT/p_${W}_${C} $C $W <<"PY" >> $OUT/pmc_small.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "small_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
g = max(int(r["Grid_Size"]) for r in rows)
full = [r for r in rows if int(r["Grid_Size"]) == g]
print(sys.argv[3], sys.argv[2], "KB, last full-batch launch:", full[-1]["Counter_Value"], full[-1]["Kernel_Name"][:40], "launches", len(full))
PY
  rm -rf $OUT/p_${W}_${C}
done
done
cat $OUT/pmc_small.txt
