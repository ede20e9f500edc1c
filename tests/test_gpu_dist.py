"""-m gpu: the N > 1 path on the one GPU of the box.  Two rank processes share the card (the gloo rehearsal,
ASLAM_DIST_BACKEND=gloo: collectives staged through host memory), each owns its block of trajectories, and the gathered pose
streams must match the CPU oracle in GLOBAL trajectory order.  Also `bench.py --gpus 2` started plainly (it spawns its ranks
itself) in weak and in strong (configs[4]-shaped) mode.  SURVEY.md 8(e)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from awesomeslam_amd import dist as adist
    from awesomeslam_amd import trace as tg
    from awesomeslam_amd.core import Core
    rank, world, local = adist.init()
    assert world == 2
    torch.cuda.set_device(0)
    L, T, B = 8, 120, 2
    tr = tg.make_traces(L, T, B=B, seed=5, first_traj=rank * B)
    core = Core("ekf", tg.dim_cap(L), batch=B, max_obs=tr.max_obs, max_wait=256)
    core.set_trace(tr)
    poses = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    core.replay(0, T, poses.data_ptr(), None)
    torch.cuda.synchronize()
    adist.barrier()
    allp = adist.gather_poses(poses).cpu().numpy()
    assert allp.shape == (world * B, T, 3)
    if rank == 0:
        from oracle.c_oracle import CFilter
        ref = tg.make_traces(L, T, B=world * B, seed=5)
        for g in range(world * B):
            po, _ = CFilter("ekf", tg.dim_cap(L)).replay(ref[g])
            err = np.abs(allp[g] - po).max() / np.abs(po).max()
            assert err < 1e-6, (g, err)
            # and it is THIS trajectory's stream, not a neighbour's
            others = [np.abs(allp[h] - po).max() for h in range(world * B) if h != g]
            assert min(others) > 1e-3, (g, others)
    adist.barrier()
    adist.finalize()
    print("rank", rank, "ok")
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_share_the_gpu_and_gather_in_global_order(tmp_path, built):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ASLAM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()
        assert f"rank {r} ok" in out.decode()


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(mode, built):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts two fresh rank processes and relays rank 0's line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ASLAM_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ekf8", "--steps", "2", "--warmup", "1",
           "--chunk", "50", "--cpu-sample", "0"]
    cmd += ["--batch", "3"] if mode == "weak" else ["--scaling", "strong", "--trajectories", "4"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == mode
    per, total = (3, 6) if mode == "weak" else (2, 4)
    assert line["config"]["trajectories_per_gpu"] == per and line["config"]["trajectories_total"] == total
    assert line["value"] > 0 and abs(line["value"] - total * 50 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
