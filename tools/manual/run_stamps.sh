set -e
timeout -k 10 100 python tools/gpu_stamps.py 2>&1 | grep -v amdgpu.ids | head -14
