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
    from awesomeslam_amd.core import Core, F32, F64
    rank, world, local = adist.init()
    assert world == 2
    torch.cuda.set_device(0)
    L, T, B, f32, tol = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "f32", float(sys.argv[5])
    tr = tg.make_traces(L, T, B=B, seed=5, first_traj=rank * B)
    core = Core("ekf", tg.dim_cap(L), batch=B, max_obs=tr.max_obs, max_wait=min(2048, 2 * L + 240), dtype=F32 if f32 else F64)
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
            print("trajectory", g, "pose stream rel err vs oracle", err)
            assert err < tol, (g, err)
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


# (landmarks, callbacks, trajectories per rank, dtype, pose-stream bar).  The second case is BASELINE configs[4]'s workload on two ranks:
# EKF, 512 landmarks (n = 1027), binary32 products, ONE trajectory per rank (the batch-1 launch chain), 42 callbacks = the three growth
# stages + the first steady-state callbacks (what the CPU oracle replays in about a minute); the pose stream is a state output: 1e-6.
@pytest.mark.parametrize("L,T,B,dtype,tol", [(8, 120, 2, "f64", 1e-6), (512, 42, 1, "f32", 1e-6)])
def test_two_ranks_share_the_gpu_and_gather_in_global_order(L, T, B, dtype, tol, tmp_path, built):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ASLAM_DIST_BACKEND="gloo")
        env.pop("ASLAM_CHOL_RESIDENT", None)
        procs.append(subprocess.Popen([sys.executable, str(script), str(L), str(T), str(B), dtype, str(tol)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()
        assert f"rank {r} ok" in out.decode()
        if r == 0:
            print(out.decode())


@pytest.mark.parametrize("mode", ["weak", "strong", "configs4"])
def test_bench_spawns_its_own_ranks(mode, built):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts two fresh rank processes and relays rank 0's line.
    "configs4": BASELINE configs[4] as written, on two ranks -- `--workload ekf512 --scaling strong`, one 512-landmark fp32 filter per
    rank --, whose timed region ends with the bench's own check of the fp32 filters against the fp64 path (`parity_check`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ASLAM_CHOL_RESIDENT")}
    env["ASLAM_DIST_BACKEND"] = "gloo"
    wl, chunk = ("ekf512", 5) if mode == "configs4" else ("ekf8", 50)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", wl, "--steps", "2", "--warmup", "1",
           "--chunk", str(chunk), "--cpu-sample", "0"]
    cmd += ["--batch", "3"] if mode == "weak" else ["--scaling", "strong", "--trajectories", "2" if mode == "configs4" else "4"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == ("weak" if mode == "weak" else "strong")
    per, total = {"weak": (3, 6), "strong": (2, 4), "configs4": (1, 2)}[mode]
    assert line["config"]["trajectories_per_gpu"] == per and line["config"]["trajectories_total"] == total
    assert line["value"] > 0 and abs(line["value"] - total * chunk * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    if mode == "configs4":
        assert line["config"]["state_dim"] == 1027 and line["dtype"] == "f32" and "configs[4]" in line["config"]["workload"]
        assert line["parity_check"]["ok"], line["parity_check"]
