"""manual: one fp32 slam() of the large path at state sizes that fill 9 .. 17 block columns, against the oracle"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_large import synth
from util import rel_err, cov_err
from awesomeslam_amd.core import Core, F32
from oracle.c_oracle import CFilter
for nbk in (9, 10, 12, 14, 15, 16, 17):
    n = 64 * nbk - 1 - (0 if nbk < 17 else 0)
    if n % 2 == 0:
        n -= 1
    X, Z, P = synth(n, n)
    o = CFilter("ekf", n + 1); o.set_state(n, X, Z, P, 0.07, -0.03)
    core = Core("ekf", n + 1, batch=2, max_obs=4, max_wait=4, dtype=F32)
    core.set_state(1, n, X, Z, P)
    Xg = core.ekf_step(1, 0.2, 0.1, 1.0, Z, 0.07, -0.03); o.slam(0.2, 0.1, 1.0)
    Xo, _, Po = o.state(); Pg = core.state(1)[2]
    print(f"n={n} blocks={nbk}: rel err X {rel_err(Xg, Xo):.2e} P {cov_err(Pg, Po):.2e}", flush=True)
    core.close()
