"""Long-horizon evidence for BASELINE configs (2) and (4); writes a text report to stdout.

  part 1  EKF, 512 landmarks: the fp32 large path against the fp64 large path (itself within 1e-14 of the oracle,
          tests/test_gpu_large.py) on the same trace, relative error of pose / X / P at checkpoints
  part 2  EKF, 64 landmarks, fp64: 100 000 callbacks for 64 filters (status, finiteness, pose error against the simulated
          truth), and parity of trajectory 0 against the CPU oracle over the first 20 000 callbacks

    python tests/manual/long_run.py [--t512 20000] [--t64 100000] [--oracle 20000]
"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import awesomeslam_amd.trace as tg
from awesomeslam_amd.core import Core, F32, F64


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def part1(T, B=2, L=512, chunk=1000):
    print(f"== part 1: EKF L={L} (n={tg.full_dim(L)}), fp32 vs fp64 large path, {B} trajectories, {T} callbacks", flush=True)
    tr = tg.make_traces(L, T, B=B, seed=4)
    cores = {}
    for name, dt in (("f32", F32), ("f64", F64)):
        c = Core("ekf", tg.dim_cap(L), batch=B, max_obs=tr.max_obs, max_wait=2048, dtype=dt)
        c.set_trace(tr)
        cores[name] = c
    poses = {k: torch.zeros((B, chunk, 3), dtype=torch.float64, device="cuda") for k in cores}
    print("callbacks   N    rel err pose(chunk)   X          P         max|P32 - P32^T|/max|P|   status32", flush=True)
    marks = {1000, 2000, 5000, 10000, 20000, 50000, 100000}
    for t0 in range(0, T, chunk):
        m = min(chunk, T - t0)
        for k, c in cores.items():
            c.replay(t0, m, poses[k].data_ptr(), None)
        torch.cuda.synchronize()
        if (t0 + m) in marks or t0 + m == T:
            X32, _, P32 = cores["f32"].state(0)
            X64, _, P64 = cores["f64"].state(0)
            asym = float(np.abs(P32 - P32.T).max() / np.abs(P32).max())
            print(f"{t0 + m:9d} {cores['f32'].dim(0):4d}    {rel(poses['f32'].cpu().numpy()[0, :m], poses['f64'].cpu().numpy()[0, :m]):.3e}"
                  f"          {rel(X32, X64):.3e}  {rel(P32, P64):.3e}  {asym:.3e}                 {cores['f32'].status(0)}", flush=True)


def part2(T, T_or, B=64, L=64, chunk=2000):
    print(f"== part 2: EKF L={L} (n={tg.full_dim(L)}) fp64, {B} filters, {T} callbacks", flush=True)
    t = time.time()
    tr = tg.make_traces(L, T, B=8, seed=2)  # 8 distinct trajectories, tiled to the batch (host memory)
    reps = B // tr.B
    big = tg.Trace(*(np.concatenate([getattr(tr, f)] * reps) for f in ('odom', 'dt', 'obs_new', 'n_obs', 'obs', 'landmarks', 'truth')), tr.warmup)
    print(f"   trace generated in {time.time() - t:.0f} s", flush=True)
    c = Core("ekf", tg.dim_cap(L), batch=B, max_obs=tr.max_obs, max_wait=512)
    c.set_trace(big)
    poses = torch.zeros((B, chunk, 3), dtype=torch.float64, device="cuda")
    keep0 = []
    t = time.time()
    worst = 0.0
    for t0 in range(0, T, chunk):
        m = min(chunk, T - t0)
        c.replay(t0, m, poses.data_ptr(), None)
        torch.cuda.synchronize()
        p = poses.cpu().numpy()[:, :m]
        if not np.isfinite(p).all():
            print(f"   NON-FINITE pose in callbacks {t0}..{t0 + m}")
            return
        if t0 >= 64:
            truth = big.truth[:, t0:t0 + m, :2]
            worst = max(worst, float(np.abs(p[:, :, :2] - truth).max()))
        if t0 < T_or:
            keep0.append(p[0].copy())
    el = time.time() - t
    st = [c.status(b) for b in range(B)]
    dm = [c.dim(b) for b in range(B)]
    print(f"   {T} callbacks x {B} filters in {el:.1f} s wall ({T * B / el:.3e} filter-steps/s incl. per-chunk pose copies)")
    print(f"   status bits: {sorted(set(st))}, dimensions: {sorted(set(dm))}, max |pose - truth| (x, y) after warm-up: {worst:.3f} m")
    if T_or > 0:
        from oracle.c_oracle import CFilter
        o = CFilter("ekf", tg.dim_cap(L))
        t = time.time()
        po, _ = o.replay(tr[0].slice(0, T_or))
        print(f"   oracle: {T_or} callbacks of trajectory 0 in {time.time() - t:.0f} s")
        pd = np.concatenate(keep0)[:T_or]
        for a in (1000, 5000, 10000, 20000, 50000):
            if a <= T_or:
                print(f"   parity vs oracle, callbacks 0..{a}: rel err pose = {rel(pd[:a], po[:a]):.3e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--t512", type=int, default=20000)
    ap.add_argument("--t64", type=int, default=100000)
    ap.add_argument("--oracle", type=int, default=20000)
    a = ap.parse_args()
    if a.t512 > 0:
        part1(a.t512)
    if a.t64 > 0:
        part2(a.t64, a.oracle)
