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


def test_traffic_is_reported_only_for_the_kernel_it_was_measured_on():
    """roofline.traffic comes from a stored PMC measurement (profiles/pmc_traffic.json): it must be null, with a note, when the entry's
    "kernel" is not the chain this run launched (round-3 verdict: a changed chain must not report stale bytes silently)."""
    import json

    import bench

    ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["ekf64"]
    r = bench.roofline("ekf64", 256, 500, 0.035, ent["kernel"])
    assert r["traffic"] == ent["hbm_bytes_per_launch"] and r["traffic_note"] is None and r["hbm"] is not None
    r = bench.roofline("ekf64", 128, 250, 0.035, ent["kernel"])
    assert r["traffic"] == ent["hbm_bytes_per_launch"] / 4  # scaled to the launch
    r = bench.roofline("ekf64", 256, 500, 0.035, "some_other_kernel<9,0>")
    assert r["traffic"] is None and r["hbm"] is None and "some_other_kernel" in r["traffic_note"]


def test_more_ranks_than_gpus_is_refused(monkeypatch):
    """`--gpus N` beyond torch.cuda.device_count() stops with a clear message before any rank is spawned (the first RCCL run is the
    driver's: two ranks on one device would hang in the communicator set-up); ASLAM_DIST_BACKEND=gloo is the rehearsal that shares devices."""
    import bench

    monkeypatch.delenv("ASLAM_DIST_BACKEND", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: pytest.fail("ranks were spawned"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "this node has 1 GPU(s)" in str(e.value)
