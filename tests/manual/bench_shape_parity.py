"""Manual GPU diagnostic: the fp32 large path at bench.py's shape (256 filters, seed 1, 64 + 12 x 20 callbacks) against the fp64 large path,
per filter and per covariance block.  python tests/manual/bench_shape_parity.py [B] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from awesomeslam_amd import trace as tg
from awesomeslam_amd.core import Core, F32, F64
from util import block_rel_err, rel_err

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 304
L = 512
ids = sorted({0, 3, min(B - 1, 77), min(B - 1, 130), min(B - 1, 200), B - 1})
tr = tg.make_traces(L, T, B=B, seed=1)
sub = tr.select(ids)
main = Core("ekf", tg.dim_cap(L), batch=B, max_obs=tr.max_obs, max_wait=min(2048, 2 * L + 64), dtype=F32)
main.set_trace(tr)
ref = Core("ekf", tg.dim_cap(L), batch=len(ids), max_obs=tr.max_obs, max_wait=min(2048, 2 * L + 64), dtype=F64)
ref.set_trace(sub)
t = 0
for chunk in [64] + [20] * ((T - 64) // 20):
    main.replay(t, chunk, None, None)
    ref.replay(t, chunk, None, None)
    t += chunk
    torch.cuda.synchronize()
    if t in (64, 104, 184, 304) or t == T:
        for i, b in enumerate(ids):
            X, Z, P = main.state(b)
            Xr, Zr, Pr = ref.state(i)
            eb = block_rel_err(P, Pr)
            print(f"t={t:4d} filter {b:3d} N={X.shape[0]} Z equal {np.array_equal(Z, Zr)}  X {rel_err(X, Xr):.2e} (pose {rel_err(X[:3], Xr[:3]):.2e})  P {rel_err(P, Pr):.2e}  "
                  f"blocks pose/cross/landmark {eb[0]:.2e} {eb[1]:.2e} {eb[2]:.2e}  status {main.status(b)} launch {main.launch_info()}", flush=True)
            if t == T and rel_err(P, Pr) > 8e-7:
                d = np.abs(P - Pr)
                i0, j0 = np.unravel_index(np.argmax(d), d.shape)
                print(f"      worst entry ({i0},{j0}): {P[i0, j0]:.9e} vs {Pr[i0, j0]:.9e}; max|P| {np.abs(Pr).max():.3e}; X worst at {int(np.argmax(np.abs(X - Xr)))}")
