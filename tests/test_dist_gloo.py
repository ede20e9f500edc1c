"""The N > 1 path on CPU: world_size-2 gloo processes shard trajectories and gather poses (no GPU needed)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import torch
    from awesomeslam_amd import dist as adist
    rank, world, local = adist.init("gloo")
    assert world == 2
    B, T = 3, 5
    own = list(adist.shard_range(B, rank))
    assert own == [rank * B + i for i in range(B)]
    # a pose stream that encodes (global trajectory, callback)
    poses = torch.tensor([[[g, t, g * 100 + t] for t in range(T)] for g in own], dtype=torch.float64)
    adist.barrier()
    allp = adist.gather_poses(poses)
    assert allp.shape == (world * B, T, 3)
    for g in range(world * B):
        assert allp[g, :, 0].eq(g).all() and allp[g, 3, 2].item() == g * 100 + 3
    assert adist.max_over_ranks(1.0 + rank) == 2.0
    assert adist.sum_over_ranks(1.0 + rank) == 3.0
    adist.finalize()
    print("rank", rank, "ok")
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world2_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()
        assert f"rank {r} ok" in out.decode()


def test_single_process_is_a_no_op():
    import torch
    from awesomeslam_amd import dist as adist

    x = torch.zeros((2, 4, 3), dtype=torch.float64)
    assert adist.gather_poses(x) is x
    assert adist.max_over_ranks(3.5) == 3.5
    assert list(adist.shard_range(4, 2)) == [8, 9, 10, 11]
