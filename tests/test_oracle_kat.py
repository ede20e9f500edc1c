"""Known-answer tests of the oracle's scalar pieces.

The reference has no tests; the expected values below are the binary32 facts probed
for SURVEY.md F3 and values derived by hand from the reference text (tools.h:44-66, ukf.h:73-81,
config.h:39-65)."""
import math

import numpy as np
import pytest

from oracle import c_oracle, np_oracle


def test_pi_is_binary32():
    # const float PI = 3.141592654 (config.h:39)
    assert float(np_oracle.PI) == 3.1415927410125732
    assert float(np_oracle.TWO_PI) == 6.2831854820251465


@pytest.mark.parametrize("theta", [0.0, 1.0, -1.0, 3.0, 3.2, -3.2, 6.0, -6.0, 6.4, -6.4, 9.5, -9.5, 100.0, -100.0,
                                   3.1415927410125732, -3.1415927410125732, 3.1415929794311523, 6.2831854820251465,
                                   1e-30, 12.566370964050293])
def test_normalize_angle_agree_and_range(theta, built):
    a = c_oracle.normalize_angle(theta)
    b = np_oracle.normalize_angle(theta)
    assert a == b
    assert -float(np_oracle.PI) <= float(a) <= float(np_oracle.PI)
    # idempotent: the nodes re-normalise a stored bearing on every re-walk (ekf.cpp:147-150)
    assert c_oracle.normalize_angle(a) == a
    # congruent to theta modulo float(2 pi), up to binary32 rounding of the wrap
    k = round((float(np.float32(theta)) - float(a)) / float(np_oracle.TWO_PI))
    assert abs(float(np.float32(theta)) - float(a) - k * float(np_oracle.TWO_PI)) < 1e-4


def test_normalize_angle_values(built):
    assert float(c_oracle.normalize_angle(3.2)) == pytest.approx(3.2 - 6.2831854820251465, abs=3e-7)
    assert float(c_oracle.normalize_angle(-3.2)) == pytest.approx(-3.2 + 6.2831854820251465, abs=3e-7)
    assert float(c_oracle.normalize_angle(1.5)) == float(np.float32(1.5))


@pytest.mark.parametrize("yaw", [0.0, 0.3, -0.3, 1.5707963, 3.0, -3.0, 3.14159])
def test_quat2euler(yaw, built):
    w, z = math.cos(yaw / 2), math.sin(yaw / 2)
    a = c_oracle.quat2euler(w, 0.0, 0.0, z)
    b = np_oracle.quat2euler(w, 0.0, 0.0, z)
    assert a == b
    assert float(a) == pytest.approx(yaw, abs=2e-6)


@pytest.mark.parametrize("N,wsum", [(3, None), (13, None), (131, 1.0000025928)])
def test_ukf_weights(N, wsum, built):
    f = c_oracle.CFilter("ukf", 400)
    X = np.zeros(N)
    f.set_state(N, X, X, np.eye(N))
    w, lam = f.weights()
    g = np_oracle.NpFilter("ukf", 400)
    g.update_weights(N)
    assert np.array_equal(w, g.weights) and lam == g.lam
    assert float(lam) == 3.0 - (N + 2)
    assert w[1] == 0.1666666716337204  # float(0.5 / 3)
    assert np.all(w[1:] == w[1]) and len(w) == 2 * N + 5
    assert w[0] == float(np.float32(np.float32(lam) / np.float32(3.0)))
    if wsum is not None:
        assert float(w.sum()) == pytest.approx(wsum, abs=1e-9)  # the weights do NOT sum to 1


def test_initial_state(built):
    for kind in ("ekf", "ukf"):
        f = c_oracle.CFilter(kind, 30)
        X, Z, P = f.state()
        assert f.N == 3 and np.all(X == 0) and np.all(Z == 0)
        assert np.array_equal(P, np.eye(3) * float(np.float32(0.001)))
    assert c_oracle.CFilter("ekf", 30).A() == (1.0, 0.0)
