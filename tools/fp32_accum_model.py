"""CPU model of the ACCUMULATION ORDER of the large path's binary32 chain (round 3): where does the 2e-7 .. 1.3e-6 of the fp32
covariance come from, and which cheap changes of the kernels remove it?

The chain of ekf_large*.h is modelled block by block (64-wide block columns, left-looking by block row like large_chol_resident and
the strip sweep of large_trsm_pipe), every partial product formed in binary32 and added to a binary32 accumulator every `chunk`
columns of the contraction -- the roundings an MFMA accumulator tile sees.  Variants:
    zero      accumulators start at 0, history blocks j = 0 .. k-1, then C = G - acc          (the round-2 kernels)
    init      accumulators start at -G (or -S): the partial sums are RESIDUALS, small once the dominant leading columns are in
    syrk_rev  P -= V V^T accumulated from the last K slab to the first (small terms first)
    syrk_blk  P -= V V^T: the fp32 accumulator is flushed into the fp64 P every `flush` columns of the contraction
TEST/DESIGN TOOL: imports oracle/, never imported by the product.

    python tools/fp32_accum_model.py [landmarks] [callbacks] [seed]
"""
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from awesomeslam_amd import trace as tg  # noqa: E402
from oracle.np_oracle import NpFilter, measurement, normalize_angle, state_transition  # noqa: E402

f32 = np.float32
LB = 64


def acc_chain(acc, A, B, chunk, reverse=False):
    """acc + A B^T with the contraction visited in chunks, one binary32 rounding of the accumulator per chunk"""
    K = A.shape[1]
    starts = list(range(0, K, chunk))
    if reverse:
        starts = starts[::-1]
    for s in starts:
        acc = (acc + A[:, s:s + chunk] @ B[:, s:s + chunk].T).astype(f32)
    return acc


class Chain(NpFilter):
    def __init__(self, cap, init=False, syrk="fwd", chunk=4, syrk_chunk=32, flush=0, exact=()):
        super().__init__("ekf", cap)
        self.init, self.syrk, self.chunk, self.syrk_chunk, self.flush, self.exact = init, syrk, chunk, syrk_chunk, flush, set(exact)

    def sweep(self, Rows, Lm, Linv, nbk, chol_diag=False):
        """rows of G (or block row nbk of S) -> X = Rows L^-T over block columns 0 .. nbk-1; with chol_diag also C of column nbk"""
        m = Rows.shape[0]
        X = np.zeros((m, LB * (nbk + 1)), f32)
        for k in range(nbk + (1 if chol_diag else 0)):
            Gk = Rows[:, LB * k:LB * k + LB]
            acc = (-Gk).astype(f32) if self.init else np.zeros_like(Gk)
            hist = Lm[LB * k:LB * k + LB, :LB * k] if not (chol_diag and k == nbk) else X[:, :LB * k]
            if k > 0:
                acc = acc_chain(acc, X[:, :LB * k], hist, self.chunk)
            C = (-acc) if self.init else (Gk - acc).astype(f32)
            if chol_diag and k == nbk:
                return X, C
            X[:, LB * k:LB * k + LB] = acc_chain(np.zeros_like(C), C, Linv[k], self.chunk)
        return X, None

    def _slam_ekf(self, vx, az, dt):
        N = self.N
        self.X = state_transition(N, self.X, vx, az, dt)
        self.X[2] = float(normalize_angle(self.X[2]))
        P = self.A @ self.P @ self.A.T + self.Q
        self._update_h()
        G64 = P @ self.H.T
        S64 = self.H @ G64 + self.R
        Y = self.Z - measurement(N, self.X)
        self._wrap_even(Y)
        NP = (N + 1 + LB - 1) // LB * LB
        nb = NP // LB
        G = np.zeros((NP, NP), f32)
        G[:N, :N] = G64.astype(f32)
        G[N, :N] = Y.astype(f32)
        S = np.eye(NP, dtype=f32)
        S[:N, :N] = np.tril(S64).astype(f32)
        if "chol" in self.exact:
            Lm = np.zeros((NP, NP), f32)
            Lm[:N, :N] = np.linalg.cholesky(S64).astype(f32)
            Lm[N:, N:] = np.eye(NP - N, dtype=f32)
            Linv = [sl.solve_triangular(Lm[LB * k:LB * k + LB, LB * k:LB * k + LB].astype(np.float64), np.eye(LB), lower=True).astype(f32) for k in range(nb)]
        else:
            Lm = np.zeros((NP, NP), f32)
            Linv = []
            for I in range(nb):
                X, C = self.sweep(S[LB * I:LB * I + LB, :], Lm, Linv, I, chol_diag=True)
                Lm[LB * I:LB * I + LB, :LB * I] = X[:, :LB * I]
                Cs = np.tril(C.astype(np.float64))
                Cs = Cs + np.tril(Cs, -1).T
                Ld = np.linalg.cholesky(Cs)  # diagonal block: fp64 tiles on the device
                Lm[LB * I:LB * I + LB, LB * I:LB * I + LB] = Ld.astype(f32)
                Linv.append(sl.solve_triangular(Ld, np.eye(LB), lower=True).astype(f32))
        if "trsm" in self.exact:
            V = sl.solve_triangular(Lm.astype(np.float64), G.astype(np.float64).T, lower=True).T.astype(f32)
        else:
            V, _ = self.sweep(G, Lm, Linv, nb)
        q = V[N, :].astype(np.float64)
        Vn = V[:N, :]
        self.X = self.X + Vn.astype(np.float64) @ q
        if "syrk" in self.exact:
            Pn = P - Vn.astype(np.float64) @ Vn.astype(np.float64).T
        elif self.flush:
            Pn = P.copy()
            for s in range(0, NP, self.flush):
                e = min(s + self.flush, NP)
                Pn -= acc_chain(np.zeros((N, N), f32), Vn[:, s:e], Vn[:, s:e], self.syrk_chunk).astype(np.float64)
        else:
            Pn = P - acc_chain(np.zeros((N, N), f32), Vn, Vn, self.syrk_chunk, reverse=(self.syrk == "rev")).astype(np.float64)
        self.P = np.tril(Pn) + np.tril(Pn, -1).T


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def blockwise(P, Po):
    return rel(P[:3, :3], Po[:3, :3]), rel(P[3:, :3], Po[3:, :3]), rel(P[3:, 3:], Po[3:, 3:])


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    tr = tg.make_traces(L, T, B=1, seed=seed)[0]
    cap = tg.dim_cap(L)
    filt = {
        "ref": NpFilter("ekf", cap),
        "zero": Chain(cap),
        "init": Chain(cap, init=True),
        "init+rev": Chain(cap, init=True, syrk="rev"),
        "init+fl256": Chain(cap, init=True, flush=256),
        "zero,xchol": Chain(cap, exact=("chol",)),
        "zero,xtrsm": Chain(cap, exact=("trsm",)),
        "zero,xsyrk": Chain(cap, exact=("syrk",)),
        "xall": Chain(cap, exact=("chol", "trsm", "syrk")),
    }
    marks = sorted(set(list(range(20, T + 1, 20)) + [T]))
    print(f"EKF, {L} landmarks, {T} callbacks, seed {seed}; relative error vs the fp64 oracle: P norm-wise (pose / cross / landmark block)")
    for t in range(T):
        for f in filt.values():
            if tr.obs_new[t]:
                k = int(tr.n_obs[t])
                f.sensor_msg(tr.obs[t, :k, 0], tr.obs[t, :k, 1])
            f.odom_msg(*tr.odom[t], tr.dt[t])
        if t + 1 in marks:
            ref = filt["ref"]
            for name, f in filt.items():
                if name == "ref":
                    continue
                assert f.N == ref.N
                bw = blockwise(f.P, ref.P)
                print(f"  t={t + 1:5d} N={f.N:4d} {name:12s} X {rel(f.X, ref.X):.2e}  P {rel(f.P, ref.P):.2e}  ({bw[0]:.2e} {bw[1]:.2e} {bw[2]:.2e})", flush=True)


if __name__ == "__main__":
    main()
