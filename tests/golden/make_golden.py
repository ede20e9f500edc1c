"""Generate the committed golden fixtures from the CPU oracles.

The reference (iamarkaj/AwesomeSLAM) ships no tests, fixtures or golden vectors and cannot be built in this
image (no Eigen, no ROS), so these vectors are SELF-GENERATED: they freeze the outputs of the two
independent restatements (oracle/aslam_oracle.cpp and oracle/np_oracle.py, which must agree to 1e-9 here)
on small seeded traces.  A fixture is data only: the input trace and the expected outputs.

    python tests/golden/make_golden.py        (re-run only when the oracle or the generator changes on purpose)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from awesomeslam_amd import trace as tg  # noqa: E402
from oracle.c_oracle import CFilter  # noqa: E402
from oracle.np_oracle import NpFilter  # noqa: E402

# name -> (filter, L, T, generator kwargs, max_landmark_count or None for "just enough")
CASES = {
    "ekf_L5": ("ekf", 5, 400, dict(seed=101), 30),        # BASELINE config 1 as named (5 landmarks), shipped cap 30
    "ekf_L8": ("ekf", 8, 400, dict(seed=102), 30),        # config 1 as shipped (landmarks.yaml has 8), cap 30
    "ekf_L13_cap": ("ekf", 14, 100, dict(seed=103, stages=2), 30),  # 14 landmarks against the shipped cap: last growth refused
    "ekf_L8_rewalk": ("ekf", 8, 300, dict(seed=104, sensor_every=3, dt_mode="random"), 30),
    "ekf_L8_junk": ("ekf", 8, 300, dict(seed=105, warm_hop=12, layout="ring", sensor_range=6.0), 30),
    "ukf_L5": ("ukf", 5, 300, dict(seed=111), 30),
    "ukf_L8": ("ukf", 8, 300, dict(seed=112), 30),
    "ukf_L8_rewalk": ("ukf", 8, 200, dict(seed=113, sensor_every=2, dt_mode="random"), 30),
    "ekf_L64": ("ekf", 64, 120, dict(seed=121), None),    # config 2 geometry, a short prefix
    "ukf_L64": ("ukf", 64, 80, dict(seed=122), None),     # config 3 geometry, a short prefix
}


def run_case(name):
    kind, L, T, kw, cap = CASES[name]
    cap = tg.dim_cap(L) if cap is None else cap
    tr = tg.make_traces(L, T, B=1, **kw)
    c = CFilter(kind, cap)
    pc, dc = c.replay(tr[0])
    n = NpFilter(kind, cap)
    pn, dn = n.replay(tr[0])
    Xc, Zc, Pc = c.state()
    assert np.array_equal(dc, dn), name
    assert np.array_equal(Zc, n.Z), name
    for a, b in ((pc, pn), (Xc, n.X), (Pc, n.P)):
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max()), (name, np.abs(a - b).max())
    wr, wb, wc = c.wait_list()
    return dict(kind=kind, L=L, cap=cap, odom=tr.odom[0], dt=tr.dt[0], obs_new=tr.obs_new[0], n_obs=tr.n_obs[0],
                obs=tr.obs[0], poses=pc, dims=dc, X=Xc, Z=Zc, P=Pc, wait_range=wr, wait_bearing=wb, wait_count=wc,
                A=np.array(c.A()))


if __name__ == "__main__":
    for name in CASES:
        out = run_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "N =", out["dims"][-1], "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))
