# Profiling passes of the round's final chain (configs[3]: EKF, 512 landmarks, fp32 products, 256 filters).  Run on the GPU box through gpurun:
#   gpurun --timeout 900 -- 'bash tools/manual/profile_round.sh r3p'
# Outputs under gpurun_out/<tag>/ (copy what is to be kept into profiles/):
#   stats4/          rocprofv3 --kernel-trace --stats of the default launch shape (4 stream groups)
#   kernel_stats_1group.csv, one_stream_breakdown.txt   kernel trace + stats with ONE stream group (every launch covers all 256 filters: single-stream kernel durations)
#   sq/              --pmc SQ_* pass (one stream group): MFMA utilisation per kernel (tools/pmc_summary.py)
#   fetch/, write/   --pmc FETCH_SIZE / WRITE_SIZE passes (separate, as the guide prescribes)
# rocprofv3 gets the program itself after `--` (python3 bench.py ...): no env / bash -c hop (the box refuses an exec from a process that holds the GPU).
TAG=${1:-r3p}
OUT=gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --no-sub --no-legs --cpu-sample 0 --steps 5 --warmup 2"   # (the synthetic trace depends on its length: with --steps 3 one landmark of trajectory 224 is not promoted in the prologue -- in the CPU oracle too -- and bench.py refuses to time that)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -o s -- $BENCH > $OUT/stats4_bench.json 2> $OUT/stats4.err
export ASLAM_LARGE_GROUPS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- $BENCH > $OUT/trace1_bench.json 2> $OUT/trace1.err
find $OUT/trace1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_1group.csv \;
python3 - $OUT <<"PY"
import csv, collections, sys, glob
out = sys.argv[1]
f = glob.glob(out + "/trace1/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grid = lambda r: int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
rows = [r for r in rows if "<double" not in r["Kernel_Name"]]  # (bench.py's parity check replays three filters through the fp64 path afterwards)
fe = [r for r in rows if "frontend" in r["Kernel_Name"] and grid(r) >= 256 * 768]
s0, s1 = int(fe[-11]["Start_Timestamp"]), int(fe[-1]["Start_Timestamp"])
agg, cnt, gap = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s0 <= s < s1:
        k = r["Kernel_Name"].split("(")[0].replace("void aslam::", "")[:40]
        agg[k] += (e - s) / 1e3
        cnt[k] += 1
        if prev_end is not None:
            gap[k] += (s - prev_end) / 1e3
    prev_end = e
with open(out + "/one_stream_breakdown.txt", "w") as o:
    print("one stream group, 256 filters: last 10 callbacks; callback period %.1f us, sum of kernel durations %.1f us" % ((s1 - s0) / 1e4, sum(agg.values()) / 10), file=o)
    for k, v in agg.most_common():
        print("   %-42s %8.1f us per callback (%d launches, %.1f us each; idle gap in front of it %.1f us)" % (k, v / 10, cnt[k] // 10, v / cnt[k], gap[k] / cnt[k]), file=o)
print(open(out + "/one_stream_breakdown.txt").read())
PY
rm -rf $OUT/trace1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 bench.py --no-sub --no-legs --cpu-sample 0 --steps 2 --warmup 2 > /dev/null 2> $OUT/sq.err
python3 tools/pmc_summary.py $OUT/sq > $OUT/pmc_sq_summary.txt 2>&1
rm -rf $OUT/sq
unset ASLAM_LARGE_GROUPS
bash tools/manual/profile_traffic.sh $TAG
cp $OUT/stats4/*/*kernel_stats.csv $OUT/kernel_stats_4groups.csv 2>/dev/null || find $OUT/stats4 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_4groups.csv \;
rm -rf $OUT/stats4
echo profile_round done
