"""bench.py's host logic without a GPU: flop models, rank spawning, and that it refuses to run without a device."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flop_models():
    import bench

    n = 1027
    alg, exe = bench.algorithmic_flops("ekf", n), bench.executed_flops("ekf512", n)
    assert abs(alg - (7.0 / 3.0) * n ** 3) < 1.0
    # padded to 17 blocks of 64, panel product instead of a triangular solve: between 1.1x and 1.5x the algorithmic figure
    assert 1.1 * alg < exe < 1.5 * alg
    assert bench.executed_flops("ekf64", 131) == 144 ** 3
    assert bench.algorithmic_flops("ukf", 131) == 10.7 * 131 ** 3


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the refusal on a box without a GPU")
def test_bench_without_gpu_fails_loudly_also_through_the_spawner():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for extra in ([], ["--gpus", "2"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ekf8", "--steps", "1", "--warmup", "0"] + extra,
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode != 0
        assert b"needs a GPU" in r.stderr
        assert b'"metric"' not in r.stdout
