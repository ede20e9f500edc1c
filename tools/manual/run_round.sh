# full GPU suite + the default bench line + the batch-1 and one-stream-group breakdowns, in one box
set -e
TAG=${1:-r4}
mkdir -p gpurun_out/$TAG
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1 || { tail -30 gpurun_out/$TAG/gpu_tests.log; exit 1; }
tail -2 gpurun_out/$TAG/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
python - gpurun_out/$TAG/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ekf512", r["value"], r["roofline"]["frac"], "batch1 us", r.get("single_trajectory", {}).get("us_per_callback"))
for k, v in r.get("sub", {}).items():
    print(k, v["value"], v["roofline"]["frac"], v.get("single_trajectory", {}).get("us_per_callback"))
print("parity", r.get("parity_check"))
PY
timeout -k 10 200 bash tools/manual/profile_batch1.sh $TAG/b1 | tail -9
timeout -k 10 300 bash tools/manual/profile_quick.sh $TAG/q | tail -9
