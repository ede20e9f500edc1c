"""Helpers shared by the tests."""
import numpy as np

# north_star: "outputs within 1e-6 relative fp64".  Relative is taken norm-wise (max |a-b| / max |b|) per
# vector / matrix: individual covariance entries cross zero.  For the covariance the comparison is also made
# BLOCK-wise -- pose 3x3, pose-landmark cross block, landmark block, each against its own largest entry: the
# pose block is 10^2..10^3 times smaller than the landmark diagonal and would hide behind it in a single norm.
REL_TOL = 1e-6


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(float(np.abs(b).max()), 1e-300) if b.size else 1.0
    return float(np.abs(a - b).max() / scale) if b.size else 0.0


def block_rel_err(P, Po):
    """(pose block, cross block, landmark block) relative errors of a covariance, each against its own maximum"""
    P, Po = np.asarray(P, np.float64), np.asarray(Po, np.float64)
    n = Po.shape[0]
    if n <= 3:
        return rel_err(P, Po), 0.0, 0.0
    return rel_err(P[:3, :3], Po[:3, :3]), max(rel_err(P[3:, :3], Po[3:, :3]), rel_err(P[:3, 3:], Po[:3, 3:])), rel_err(P[3:, 3:], Po[3:, 3:])


def cov_err(P, Po):
    """worst of the norm-wise and the three block-wise relative errors"""
    return max((rel_err(P, Po),) + block_rel_err(P, Po))


def assert_parity(X, Xo, P, Po, tol=REL_TOL, what=""):
    ex, eb = rel_err(X, Xo), block_rel_err(P, Po)
    assert max((ex, rel_err(P, Po)) + eb) < tol, f"{what}: X {ex:.2e}, P {rel_err(P, Po):.2e}, P blocks pose/cross/landmark {eb[0]:.2e} {eb[1]:.2e} {eb[2]:.2e}"
    return ex, eb


def sub_trajectory(traj, t0, t1):
    return traj.slice(t0, t1)
