"""Where does the fp32 covariance error of the large path come from?  (manual diagnostic, GPU; round 3)

A realistic state (n = 3 + 2 L, after T callbacks of the fp64 large path on a synthetic trace) is stepped ONCE through the binary32 chain by the
per-callback seam, the work matrices are read back (aslam_debug_large) and every stage is replaced in turn by exact (binary64, host) arithmetic:

    total      |P_dev - P_ref| / max|P_ref|                                          what the parity tests see
    GS         the same with V, L recomputed exactly from the DEVICE's binary32 G, S  -> the cost of rounding G, S to binary32
    chol       ... exact TRSM and syrk on the device's L                              -> + the device's Cholesky
    trsm       ... exact syrk on the device's V                                       -> + the device's TRSM
    (total)    device syrk on the device's V                                          -> + the device's syrk
plus the residuals of each kernel: |L L^T - S| / |S|, |V L^T - G| / |G|, |dP_dev - V V^T| / |P|.

    python tests/manual/large_residuals.py [landmarks=512] [callbacks=120] [resident=0/1]
"""
import ctypes
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 120
os.environ["ASLAM_CHOL_RESIDENT"] = sys.argv[3] if len(sys.argv) > 3 else "0"
os.environ["ASLAM_KEEP_L32"] = "1"  # large_chol_bf16 stores the off-diagonal blocks of L in binary32 only on request (this script reads L back)

import torch  # noqa: E402,F401
from awesomeslam_amd import trace as tg  # noqa: E402
from awesomeslam_amd import core as ac  # noqa: E402
from awesomeslam_amd.core import Core, F32, F64  # noqa: E402
from oracle.np_oracle import NpFilter, measurement, normalize_angle, state_transition  # noqa: E402
from util import block_rel_err, rel_err  # noqa: E402

lib = ac.core_lib()
lib.aslam_debug_large.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int64]


def debug_large(core, traj, which, NP):
    cnt = NP * NP if which < 2 else 17 * 64 * 64 if which == 2 else NP
    out = np.empty(cnt)
    ac._chk(lib.aslam_debug_large(core._h, traj, which, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cnt))
    return out.reshape((NP, NP)) if which < 2 else out.reshape((17, 64, 64)) if which == 2 else out


def main():
    tr = tg.make_traces(L, T + 1, B=1, seed=3)
    cap = tg.dim_cap(L)
    c64 = Core("ekf", cap, batch=1, max_obs=tr.max_obs, max_wait=2048, dtype=F64)
    c64.set_trace(tr)
    c64.replay(0, T)
    c64.sync()
    X, Z, P = c64.state(0)
    a00, a10 = c64.A(0)
    n = len(X)
    print(f"state after {T} callbacks: n = {n}, max|P| = {np.abs(P).max():.3e}, pose block max {np.abs(P[:3,:3]).max():.3e}, status {c64.status(0)}")
    vx, az, dt = 0.18, 0.07, 1.0
    # reference: as-coded algebra in binary64
    o = NpFilter("ekf", cap)
    o.set_state(n, X, Z, P, a00, a10)
    o.slam(np.float32(vx), np.float32(az), np.float32(dt))
    # the chain's own quantities in binary64
    f = NpFilter("ekf", cap)
    f.set_state(n, X, Z, P, a00, a10)
    f.X = state_transition(n, f.X, vx, az, dt)
    f.X[2] = float(normalize_angle(f.X[2]))
    Pp = f.A @ f.P @ f.A.T + f.Q
    f._update_h()
    G64 = Pp @ f.H.T
    S64 = f.H @ G64 + f.R
    L64 = np.linalg.cholesky(S64)
    V64 = sl.solve_triangular(L64, G64.T, lower=True).T
    Pn64 = Pp - V64 @ V64.T
    print(f"chain algebra (V V^T form) vs as-coded reference in binary64: {rel_err(Pn64, o.P):.2e}")
    c32 = Core("ekf", cap, batch=1, max_obs=tr.max_obs, max_wait=2048, dtype=F32)
    c32.set_state(0, n, X, Z, P)
    c32.ekf_step(0, vx, az, dt, Z, a00, a10)
    Pd = c32.state(0)[2]
    NP = c32.layout()[0]
    Vd = debug_large(c32, 0, 0, NP)[:n, :]
    Ld = np.tril(debug_large(c32, 0, 1, NP))
    print("launch:", c32.launch_info(), "status", c32.status(0))
    nb = (n + 1 + 63) // 64
    na = nb * 64
    Ld = Ld[:na, :na]
    Vd = Vd[:, :na]

    def show(name, Pn):
        bw = block_rel_err(Pn, o.P)
        print(f"  {name:34s} P {rel_err(Pn, o.P):.2e}  (pose {bw[0]:.2e} cross {bw[1]:.2e} landmarks {bw[2]:.2e})")

    G32 = np.zeros((n, na))
    G32[:, :n] = G64.astype(np.float32)
    S32 = np.eye(na)
    S32[:n, :n] = S64.astype(np.float32)
    # exact chain on binary32-rounded G, S
    Lx = np.linalg.cholesky(S32)
    Vx = sl.solve_triangular(Lx, G32.T, lower=True).T
    show("G, S rounded; rest exact", Pp - Vx @ Vx.T)
    Vx = sl.solve_triangular(Ld, G32.T, lower=True).T
    show("+ device Cholesky", Pp - Vx @ Vx.T)
    show("+ device TRSM", Pp - Vd @ Vd.T)
    show("+ device syrk (= the device)", Pd)
    dP = Pp - Pd
    VVt = Vd @ Vd.T
    print(f"  residuals: |L L^T - S|/|S| {np.abs(np.tril(Ld @ Ld.T - S32)).max() / np.abs(S32).max():.2e}   "
          f"|V L^T - G|/|G| {np.abs(Vd @ Ld.T - G32).max() / np.abs(G32).max():.2e}   |dP - V V^T|/|P| {np.abs(dP - VVt).max() / np.abs(o.P).max():.2e}"
          f"   |dP|/|P| {np.abs(dP).max() / np.abs(o.P).max():.2e}")
    # is the syrk's error biased?  E > 0 means the device subtracted MORE than V V^T
    E = dP - VVt
    El, Vl = E[3:, 3:], VVt[3:, 3:]
    sg = np.sign(Vl)
    print(f"  syrk error, landmark block: mean(E sign(VV^T)) / mean|E| = {np.mean(El * sg) / np.mean(np.abs(El)):+.3f}   "
          f"diagonal: {np.mean(np.diag(El)) / np.mean(np.abs(np.diag(El))):+.3f}   mean|E| / mean|VV^T| = {np.mean(np.abs(El)) / np.mean(np.abs(Vl)):.2e}")
    RT = Vd @ Ld.T - G32
    print(f"  TRSM residual: mean(R sign(G)) / mean|R| = {np.mean(RT * np.sign(G32)) / np.mean(np.abs(RT)):+.3f}")
    # per block column: where the TRSM residual sits
    R = np.abs(Vd @ Ld.T - G32)
    print("  TRSM residual by block column (x 1e-8 of max|G|):", " ".join(f"{R[:, 64 * k:64 * k + 64].max() / np.abs(G32).max() * 1e8:.1f}" for k in range(nb)))
    RS = np.abs(np.tril(Ld @ Ld.T - S32))
    print("  Cholesky residual by block column (x 1e-8 of max|S|):", " ".join(f"{RS[:, 64 * k:64 * k + 64].max() / np.abs(S32).max() * 1e8:.1f}" for k in range(nb)))
    # magnitudes of the columns of L and V: is the leading block dominant?
    print("  max|L(:, block k)|:", " ".join(f"{np.abs(Ld[:, 64 * k:64 * k + 64]).max():.2e}" for k in range(nb)))
    print("  max|V(:, block k)|:", " ".join(f"{np.abs(Vd[:, 64 * k:64 * k + 64]).max():.2e}" for k in range(nb)))


if __name__ == "__main__":
    main()
