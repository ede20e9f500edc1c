# --pmc FETCH_SIZE / WRITE_SIZE passes over the default bench shape (separate passes, as the guide prescribes), summed per kernel over the last bench step.
#   gpurun --timeout 600 -- 'bash tools/manual/profile_traffic.sh r4t'     -> gpurun_out/<tag>/pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt
TAG=${1:-r4t}
OUT=gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
unset ASLAM_LARGE_GROUPS
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python3 bench.py --no-sub --no-legs --cpu-sample 0 --steps 2 --warmup 1 > $OUT/pmc_${C}_bench.json 2> $OUT/pmc_$C.err
  python3 - $OUT $C <<"PY"
import csv, glob, sys, collections
out, c = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/pmc_" + c + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "large_" in r["Kernel_Name"] and "<double" not in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64 * 256]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
# one bench step = 20 callbacks x 6 launches x the number of stream groups (from the name aslam_kernel_info reports in the bench line)
import json, re
line = json.loads(open(out + "/pmc_" + c + "_bench.json").read().strip().splitlines()[-1])
groups = int(re.search(r"(\d+) stream groups", line["config"]["kernel"]).group(1))
last = rows[-20 * 6 * groups:]
by = collections.Counter()
for r in last:
    by[r["Kernel_Name"].split("(")[0].replace("void aslam::", "")[:40]] += float(r["Counter_Value"])
with open(out + "/pmc_" + c + ".txt", "w") as o:
    print("%s KB over the %d dispatches of the last bench step (20 callbacks x 256 filters): %.1f" % (c, len(last), sum(by.values())), file=o)
    for k, v in by.most_common():
        print("   %-42s %14.1f KB = %6.2f MB per filter and callback" % (k, v, v / 1024 / 5120), file=o)
print(open(out + "/pmc_" + c + ".txt").read())
PY
  rm -rf $OUT/pmc_$C
done
