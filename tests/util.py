"""Helpers shared by the tests."""
import numpy as np

# north_star: "outputs within 1e-6 relative fp64".  Relative is taken norm-wise (max |a-b| / max |b|) per
# vector / matrix: individual covariance entries cross zero.
REL_TOL = 1e-6


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(float(np.abs(b).max()), 1e-300) if b.size else 1.0
    return float(np.abs(a - b).max() / scale) if b.size else 0.0


def sub_trajectory(traj, t0, t1):
    return traj.slice(t0, t1)
