OUT=gpurun_out/${1:-r4b1}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 bench.py --batch 1 --no-sub --no-legs --cpu-sample 0 --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/err.txt
python3 - $OUT <<"PY"
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/tr/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "<double" not in r["Kernel_Name"]]
fe = [i for i, r in enumerate(rows) if "frontend" in r["Kernel_Name"]]
i0, i1 = fe[-11], fe[-1]
agg, cnt, gap = collections.Counter(), collections.Counter(), collections.Counter()
prev = None
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void aslam::", "")[:44]
    agg[k] += (e - s) / 1e3; cnt[k] += 1
    if prev is not None: gap[k] += (s - prev) / 1e3
    prev = e
T = (int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e4
print("batch 1: last 10 callbacks; callback period %.1f us, sum of kernel durations %.1f us" % (T, sum(agg.values()) / 10))
for k, v in agg.most_common():
    print("   %-46s %8.1f us per callback (%5.1f launches, %.1f us each; idle gap in front %.1f us)" % (k, v / 10, cnt[k] / 10, v / cnt[k], gap[k] / cnt[k]))
PY
rm -rf $OUT/tr
