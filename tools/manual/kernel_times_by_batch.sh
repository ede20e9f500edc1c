mkdir -p gpurun_out/r03a && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 8 64 256; do
  ASLAM_LARGE_GROUPS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03a/b$B -o t -- python3 bench.py --no-sub --no-legs --cpu-sample 0 --steps 2 --warmup 2 --batch $B > /dev/null 2> gpurun_out/r03a/err_$B.txt
  python3 - $B <<"PY"
import csv, collections, sys
B = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/r03a/b%s/t_kernel_trace.csv" % B)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fe = [r for r in rows if "frontend" in r["Kernel_Name"]]
s0, s1 = int(fe[-11]["Start_Timestamp"]), int(fe[-1]["Start_Timestamp"])
agg = collections.Counter()
for r in rows:
    s = int(r["Start_Timestamp"])
    if s0 <= s < s1:
        agg[r["Kernel_Name"].split("(")[0].replace("void aslam::", "")[:30]] += (int(r["End_Timestamp"]) - s) / 1e4
print("batch", B, "one stream, us per callback:", {k: round(v, 1) for k, v in agg.items()})
PY
  rm -rf gpurun_out/r03a/b$B
done
