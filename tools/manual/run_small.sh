# the single-CU kernels' quick loop: EKF + UKF GPU tests, phase stamps (needs libaslam_core_stamps.so built with -DASLAM_STAMPS=1 -DASLAM_FE_STAMPS=1), the two sub-benchmarks
set -e
mkdir -p gpurun_out/fe
timeout -k 10 500 python -m pytest tests/test_gpu_ekf.py tests/test_gpu_ukf.py -x -q 2>&1 | tail -3
timeout -k 10 100 python tools/gpu_stamps_frontend.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 100 python tools/gpu_stamps.py 2>&1 | grep -v amdgpu.ids | head -14
timeout -k 10 100 python tools/gpu_stamps_ukf.py 2>&1 | grep -v amdgpu.ids | head -16
timeout -k 10 200 python bench.py --workload ekf64 --no-legs --cpu-sample 0 > gpurun_out/fe/ekf64.json
timeout -k 10 200 python bench.py --workload ukf64 --no-legs --cpu-sample 0 > gpurun_out/fe/ukf64.json
python - <<'PY'
import json
for w in ('ekf64','ukf64'):
    r=json.loads(open('gpurun_out/fe/%s.json'%w).read().strip().splitlines()[-1])
    print(w, r['value'], r['roofline']['frac'])
PY
