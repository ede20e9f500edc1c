#!/usr/bin/env python
"""bench.py -- filter-steps/s of the MI355X-native EKF/UKF-SLAM core on synthetic trajectories.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ekf64|ukf64|ekf8] [--batch B] [--chunk C]

One bench "step" = one replay launch = `chunk` consecutive callbacks (sensor message + odom message:
association, growth bookkeeping, predict, update) for each of the `batch` independent trajectories held by a
GPU.  The trace is already resident in HBM when the timed region starts.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank owns `batch` trajectories of its own (weak scaling) and the
only communication is one all_gather of the pose streams at the end of the timed region.
Prints ONE JSON line (rank 0).  The CPU oracle is used for the `cpu_baseline` leg and for nothing else.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from awesomeslam_amd import dist as adist  # noqa: E402
from awesomeslam_amd import trace as tg  # noqa: E402
from awesomeslam_amd.core import Core  # noqa: E402

WORKLOADS = {
    # name: (filter, landmarks, BASELINE.json config it corresponds to)
    "ekf64": ("ekf", 64, "configs[1]: EKF, 64 landmarks (n=131), fp64"),
    "ukf64": ("ukf", 64, "configs[2]: UKF, 64 landmarks (n=131), fp64"),
    "ekf8": ("ekf", 8, "configs[0] geometry on the GPU: EKF, 8 landmarks (n=19), fp64"),
    "ekf512": ("ekf", 512, "configs[3]/[4]: EKF, 512 landmarks (n=1027), fp32 covariance, multi-workgroup launch chain"),
}
# fp32 MFMA peak (MI355X_MICROARCH.md: 157.3 TFLOP/s, v_mfma_f32_16x16x4_f32 at the vector rate)
PEAK_F32_TFLOPS = 157.3
# fp64 peak: 256 CU x 4 SIMD x 16 FMA/clk x 2 x 2.4 GHz = 78.6 TFLOP/s, for v_fma_f64 and v_mfma_f64 alike
# (= half of the 157.3 TFLOP/s FP32 row of MI355X_MICROARCH.md, which lists no fp64 row of its own)
PEAK_F64_TFLOPS = 78.6
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s peak (about 6.3 TB/s achievable)


def algorithmic_flops(kind, n):
    """reference-equivalent minimum per callback."""
    return (2.0 + 1.0 / 3.0) * n ** 3 if kind == "ekf" else 10.7 * n ** 3


def cpu_baseline(kind, L, seed, prologue, sample):
    """The C++ oracle (Eigen-free restatement of the reference node, -O2, one core) on a bounded sample of the
    same workload: trajectory 0, `sample` steady-state callbacks after the warm-up prologue."""
    from oracle.cpu_bench import run

    r = run(kind, L, seed, 0, prologue, sample)
    el = r["t1"] - r["t0"]
    if L >= 256:
        what = (f"{sample} slam() calls on a synthetic state of the same dimension N={r['N']} (no warm-up: a callback takes "
                f"seconds); oracle/aslam_oracle.cpp (as-coded 18 n^3 dense algebra, fp64), g++ -O2, 1 thread, {el:.1f} s")
    else:
        what = (f"trajectory 0 of the same seed, {sample} steady-state callbacks after a {prologue}-callback "
                f"warm-up, N={r['N']}; oracle/aslam_oracle.cpp (as-coded 18 n^3 algebra), g++ -O2, 1 thread, {el:.1f} s")
    return {"value": sample / el, "unit": "filter-steps/s", "cores": 1, "kind": "port", "sample": what}


def cpu_baseline_all_cores(kind, L, seed, prologue, sample):
    """the reference node is single-threaded, so "all cores" = one trajectory per host core, one oracle
    process each (oracle/cpu_bench.py), same bounded sample per process; value = sum of the per-process rates."""
    import subprocess

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))  # a one-GPU box's CPU share is 16 cores, whatever the affinity mask says
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--kind", kind, "--landmarks", str(L), "--seed", str(seed),
           "--prologue", str(prologue), "--sample", str(sample)]
    procs = [subprocess.Popen(cmd + ["--traj", str(b)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for b in range(cores)]
    res = []
    for p_ in procs:
        out, _ = p_.communicate(timeout=600)
        if p_.returncode == 0 and out.strip():
            res.append(json.loads(out.strip().splitlines()[-1]))
    if not res:
        return None
    rate = sum(r["steps"] / (r["t1"] - r["t0"]) for r in res)
    span = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    overlap = min(r["t1"] for r in res) - max(r["t0"] for r in res)
    return {"value": rate, "unit": "filter-steps/s", "cores": len(res), "kind": "port",
            "sample": f"{len(res)} oracle processes side by side, one trajectory each (trajectories 0..{len(res) - 1} of the same "
                      f"seed), {sample} callbacks per process; value = sum of per-process rates; timed sections span {span:.1f} s "
                      f"and overlap for {max(overlap, 0.0):.1f} s"}


def single_trajectory_latency(kind, L, seed, prologue, C1, large, local, dev):
    """batch 1 is reported with every number.  One filter alone on the GPU, `C1` steady-state callbacks in one
    launch (replay seam), HIP events on the launch stream; the second of two launches is reported."""
    from awesomeslam_amd.core import F32, F64
    tr = tg.make_traces(L, prologue + 2 * C1, B=1, seed=seed)
    core = Core(kind, tg.dim_cap(L), batch=1, max_obs=tr.max_obs, max_wait=min(2048 if large else 512, 2 * L + 64), device=local,
                dtype=F32 if large else F64)
    core.set_trace(tr)
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.zeros((1, max(C1, prologue), 3), dtype=torch.float64, device=dev)
    core.replay(0, prologue, scratch.data_ptr(), None, stream)
    ms = []
    for w in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        core.replay(prologue + w * C1, C1, scratch.data_ptr(), None, stream)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ok = core.dim(0) == tg.full_dim(L) and core.status(0) == 0
    core.close()
    return {"trajectories": 1, "callbacks_per_launch": C1, "us_per_callback": ms[1] * 1e3 / C1,
            "filter_steps_per_s": C1 / (ms[1] * 1e-3), "steady_state": bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ekf64", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default 256)")
    ap.add_argument("--chunk", type=int, default=None,
                    help="callbacks per launch (= per bench step); default 500 (EKF) / 200 (UKF: the reference UKF only stays "
                         "positive definite for a few thousand callbacks at n = 131, DESIGN.md)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=None, help="callbacks timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    rank, world, local = adist.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the filter core has no CPU fallback")
    local = local % torch.cuda.device_count()  # ranks share devices only in the gloo rehearsal (ASLAM_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    kind, L, cfg_name = WORKLOADS[args.workload]
    n_full = tg.full_dim(L)
    large = args.workload == "ekf512"
    if args.chunk is None:
        args.chunk = 20 if large else 200 if kind == "ukf" else 500
    if args.batch is None:
        args.batch = 256
    B, C, K, W = args.batch, args.chunk, args.steps, args.warmup
    prologue = 64  # callbacks: the 42-callback warm-up in which the state grows to n_full, rounded up
    T = prologue + (W + K) * C

    # ---- synthetic input, distinct per trajectory and per rank, resident in HBM before timing starts
    t_gen = time.time()
    tr = tg.make_traces(L, T, B=B, seed=args.seed, first_traj=rank * B)
    t_gen = time.time() - t_gen
    from awesomeslam_amd.core import F32, F64
    core = Core(kind, tg.dim_cap(L), batch=B, max_obs=tr.max_obs, max_wait=min(2048 if large else 512, 2 * L + 64), device=local,
                dtype=F32 if large else F64)
    core.set_trace(tr)
    stream = torch.cuda.current_stream().cuda_stream
    poses = torch.zeros((K, B, C, 3), dtype=torch.float64, device=dev)
    scratch = torch.zeros((B, max(C, prologue), 3), dtype=torch.float64, device=dev)
    dims = torch.zeros((B, prologue), dtype=torch.int32, device=dev)

    core.replay(0, prologue, scratch.data_ptr(), dims.data_ptr(), stream)
    torch.cuda.synchronize()
    for b in range(0, B, max(1, B // 8)):
        if core.dim(b) != n_full or core.status(b) != 0:
            raise SystemExit(f"trajectory {b}: N={core.dim(b)} (want {n_full}), status={core.status(b)} after the warm-up")
    for w in range(W):
        core.replay(prologue + w * C, C, scratch.data_ptr(), None, stream)
    adist.gather_poses(scratch[:, :1])  # the collective once, untimed: communicator and buffers exist when the clock starts
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    adist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record()
        core.replay(prologue + (W + k) * C, C, poses[k].data_ptr(), None, stream)
        ev[k][1].record()
    all_poses = adist.gather_poses(poses.permute(1, 0, 2, 3).reshape(B, K * C, 3))  # the one collective: poses, at the end
    torch.cuda.synchronize()
    adist.barrier()
    el = time.perf_counter() - t0
    el = adist.max_over_ranks(el, dev if world > 1 else "cpu")
    launch_ms = [a.elapsed_time(b) for a, b in ev]
    kernel_s = float(np.mean(launch_ms)) * 1e-3

    ok = all(core.dim(b) == n_full and core.status(b) == 0 for b in range(0, B, max(1, B // 8)))
    finite = bool(torch.isfinite(all_poses).all().item())
    if not (ok and finite):
        raise SystemExit("bench: a filter left its steady state (dimension/status/non-finite pose)")

    if rank == 0:
        total_steps = world * B * C * K
        flops_launch = algorithmic_flops(kind, n_full) * B * C
        achieved = flops_launch / kernel_s / 1e12
        peak = PEAK_F32_TFLOPS if large else PEAK_F64_TFLOPS
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        info = core.kernel_info()
        out = {
            "metric": "EKF/UKF filter-steps/s @ N_landmarks", "value": total_steps / el, "unit": "filter-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": el / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if large else "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload} = {cfg_name}", "landmarks": L, "state_dim": n_full,
                       "trajectories_per_gpu": B, "callbacks_per_step": C, "parallelism": f"trajectory-sharded x{world}",
                       "kernel": info["name"], "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"],
                       "trace_gen_s": round(t_gen, 1)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         # the other roofline north_star asks for: PMC bytes at the L2's memory side per launch / launch time
                         "hbm": None if traffic is None else {"achieved": traffic / kernel_s / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                                              "frac": traffic / kernel_s / 1e9 / PEAK_HBM_GBPS},
                         "kernel_ms": kernel_s * 1e3,
                         "note": "achieved = SURVEY 8(d) algorithmic flops/callback (EKF 2.33 n^3, UKF 10.7 n^3: the reference-equivalent "
                                 "minimum the survey defines; the single-CU EKF kernel itself executes about n^3 since its update "
                                 "uses R = r I) x callbacks x trajectories per launch / "
                                 "mean launch duration (HIP events on the launch stream); peak = dense "
                                 + ("fp32" if large else "fp64") + " MFMA rate (MI355X_MICROARCH.md)"},
        }
        if world == 1:
            out["single_trajectory"] = single_trajectory_latency(kind, L, args.seed, prologue, min(C, 200), large, local, dev)
        sample = args.cpu_sample
        if sample is None:
            sample = {"ekf64": 1200, "ukf64": 1000, "ekf8": 20000, "ekf512": 3}[args.workload]
        if world == 1 and sample > 0:
            out["cpu_baseline"] = cpu_baseline(kind, L, args.seed, prologue, sample)
            allc = cpu_baseline_all_cores(kind, L, args.seed, prologue, sample)
            if allc:
                out["cpu_baseline_all_cores"] = allc
        print(json.dumps(out))
    adist.finalize()


if __name__ == "__main__":
    main()
