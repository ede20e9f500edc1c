set -e
timeout -k 10 100 python tools/gpu_stamps.py 2>&1 | grep -v amdgpu.ids | head -14
timeout -k 10 100 python tools/gpu_stamps_ukf.py 2>&1 | grep -v amdgpu.ids | head -40
