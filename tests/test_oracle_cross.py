"""The two independent CPU restatements (C++ and NumPy) must agree, and the algebra the filters rely on
must hold (SURVEY.md 8c pins i and iii).  "parity unpinned" by the reference: it has no tests of its own."""
import numpy as np
import pytest

from awesomeslam_amd import trace as tg
from oracle.c_oracle import CFilter
from oracle.np_oracle import NpFilter
from util import rel_err

CASES = [
    ("ekf", 5, 250, dict(seed=1), 30),
    ("ekf", 8, 250, dict(seed=2, sensor_every=3, dt_mode="random"), 30),
    ("ekf", 13, 200, dict(seed=3, stages=4), 30),
    ("ekf", 8, 200, dict(seed=4, warm_hop=12, layout="ring", sensor_range=6.0), 30),
    ("ukf", 5, 200, dict(seed=5), 30),
    ("ukf", 8, 200, dict(seed=6, sensor_every=2), 30),
    ("ukf", 13, 120, dict(seed=7), 30),
    ("ekf", 20, 80, dict(seed=8), None),
]


@pytest.mark.parametrize("kind,L,T,kw,cap", CASES)
def test_cpp_vs_numpy(kind, L, T, kw, cap, built):
    cap = tg.dim_cap(L) if cap is None else cap
    tr = tg.make_traces(L, T, B=1, **kw)[0]
    c, n = CFilter(kind, cap), NpFilter(kind, cap)
    pc, dc = c.replay(tr)
    pn, dn = n.replay(tr)
    Xc, Zc, Pc = c.state()
    assert np.array_equal(dc, dn)                      # landmark bookkeeping: bit-exact
    assert np.array_equal(Zc, n.Z)
    wr, wb, wc = c.wait_list()
    assert [float(w[0]) for w in n.wait] == [float(x) for x in wr]
    assert [float(w[1]) for w in n.wait] == [float(x) for x in wb]
    assert [int(w[2]) for w in n.wait] == [int(x) for x in wc]
    assert np.isfinite(Pc).all()
    assert rel_err(pc, pn) < 1e-10 and rel_err(Xc, n.X) < 1e-10 and rel_err(Pc, n.P) < 1e-9


def test_growth_refused_at_cap(built):
    # 14 landmarks against the shipped MAX_LANDMARK_COUNT = 30 (compared with the DIMENSION, ekf.cpp:263)
    tr = tg.make_traces(14, 120, B=1, seed=9, stages=1)[0]
    c = CFilter("ekf", 30)
    _, dims = c.replay(tr)
    assert dims[-1] == 3          # 3 + 28 = 31 >= 30: the whole batch of new landmarks is dropped
    tr = tg.make_traces(13, 120, B=1, seed=9, stages=1)[0]
    c = CFilter("ekf", 30)
    _, dims = c.replay(tr)
    assert dims[-1] == 29         # 13 landmarks is the most the shipped constant admits


def test_callbacks_before_first_sensor_message_are_dropped(built):
    tr = tg.make_traces(5, 30, B=1, seed=10)[0]
    c = CFilter("ekf", 30)
    assert c.odom_msg(*tr.odom[0], 1.0) == 0     # init_z: cbOdom returns (ekf.cpp:76)
    c.sensor_msg([], [])
    assert c.odom_msg(*tr.odom[0], 1.0) == 1     # an empty message is enough
    assert c.N == 3


@pytest.mark.parametrize("kind", ["ekf", "ukf"])
def test_algebraic_properties(kind, built):
    """S S^-1 = I, Cholesky L L^T = Paug, and for the EKF (I-KH)P = P - K S K^T on one step."""
    L, T = 8, 120
    tr = tg.make_traces(L, T, B=1, seed=12)[0]
    n = NpFilter(kind, 30)
    n.replay(tr)
    N = n.N
    Paug = np.zeros((N + 2, N + 2))
    Paug[:N, :N] = n.P
    Paug[N, N] = Paug[N + 1, N + 1] = 0.04
    Lc = np.linalg.cholesky((Paug + Paug.T) / 2)
    assert rel_err(Lc @ Lc.T, (Paug + Paug.T) / 2) < 1e-13
    assert np.linalg.eigvalsh((n.P + n.P.T) / 2).min() > 0      # the scenario keeps the oracle PD
    if kind == "ekf":
        n._update_h()
        H, P, R = n.H, n.P, n.R
        S = H @ P @ H.T + R
        assert rel_err(S @ np.linalg.inv(S), np.eye(N)) < 1e-12
        K = P @ H.T @ np.linalg.inv(S)
        assert rel_err((np.eye(N) - K @ H) @ P, P - K @ S @ K.T) < 1e-10
        # the measurement-coordinate identities the HIP kernel uses (csrc/ekf_small.h)
        Pt = H @ P @ H.T
        Kt = Pt @ np.linalg.inv(S)
        Hi = np.linalg.inv(H)
        assert rel_err(Hi @ Kt, K) < 1e-9
        assert rel_err(Hi @ (float(np.float32(0.2)) * Kt) @ Hi.T, (np.eye(N) - K @ H) @ P) < 1e-9


def test_set_state_and_single_slam(built):
    rng = np.random.default_rng(0)
    for kind in ("ekf", "ukf"):
        N = 11
        X = np.concatenate([[0.1, -0.2, 0.3], 15 + rng.normal(size=N - 3)])
        A = rng.normal(size=(N, N)) * 0.05
        P = A @ A.T + np.eye(N) * 0.01
        Z = X + rng.normal(size=N) * 0.01
        c, n = CFilter(kind, 30), NpFilter(kind, 30)
        c.set_state(N, X, Z, P, 0.05, -0.02)
        n.set_state(N, X, Z, P, 0.05, -0.02)
        for _ in range(3):
            c.slam(0.2, 0.1, 1.0)
            n.slam(np.float32(0.2), np.float32(0.1), np.float32(1.0))
        Xc, _, Pc = c.state()
        assert rel_err(Xc, n.X) < 1e-10 and rel_err(Pc, n.P) < 1e-9
