"""Both oracles against the committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py).

The fixtures are self-generated -- the reference has no golden vectors -- so what they pin is
that the oracle (and the trace generator feeding the benchmarks) does not drift."""
import glob
import os

import numpy as np
import pytest

from awesomeslam_amd.trace import Trajectory
from oracle.c_oracle import CFilter
from oracle.np_oracle import NpFilter
from util import rel_err

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    tr = Trajectory(z["odom"], z["dt"], z["obs_new"], z["n_obs"], z["obs"], None, None)
    return z, tr


def test_fixtures_present():
    assert len(GOLD) >= 10


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_cpp_oracle_matches_golden(path, built):
    z, tr = load(path)
    c = CFilter(str(z["kind"]), int(z["cap"]))
    poses, dims = c.replay(tr)
    X, Z, P = c.state()
    assert np.array_equal(dims, z["dims"]) and np.array_equal(Z, z["Z"])
    wr, wb, wc = c.wait_list()
    assert np.array_equal(wr, z["wait_range"]) and np.array_equal(wb, z["wait_bearing"]) and np.array_equal(wc, z["wait_count"])
    assert rel_err(poses, z["poses"]) < 1e-12 and rel_err(X, z["X"]) < 1e-12 and rel_err(P, z["P"]) < 1e-11


@pytest.mark.parametrize("path", [p for p in GOLD if "L64" not in p], ids=lambda p: os.path.basename(p)[:-4])
def test_numpy_oracle_matches_golden(path, built):
    z, tr = load(path)
    n = NpFilter(str(z["kind"]), int(z["cap"]))
    poses, dims = n.replay(tr)
    assert np.array_equal(dims, z["dims"]) and np.array_equal(n.Z, z["Z"])
    assert rel_err(poses, z["poses"]) < 1e-9 and rel_err(n.X, z["X"]) < 1e-9 and rel_err(n.P, z["P"]) < 1e-8
