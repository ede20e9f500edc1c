"""the scan -> (range, bearing) front end.  CPU: the oracle restatement on constructed scans;
-m gpu: the HIP kernel (include/aslam_scan.h) against the oracle."""
import numpy as np
import pytest

from oracle import scan_oracle as so


def cylinder_scan(cyls, noise=0.0, seed=0):
    """exact ray casting of 1-degree beams against circles (cx, cy, r); inf = no return"""
    rng = np.random.default_rng(seed)
    th = np.deg2rad(np.arange(360.0))
    dx, dy = np.cos(th), np.sin(th)
    best = np.full(360, np.inf)
    for cx, cy, r in cyls:
        bq = dx * cx + dy * cy
        disc = bq * bq - (cx * cx + cy * cy - r * r)
        t = np.where((disc > 0) & (bq > 0), bq - np.sqrt(np.maximum(disc, 0)), np.inf)
        best = np.minimum(best, t)
    best = np.where(np.isfinite(best), best + rng.normal(0, noise, 360), np.inf)
    return best.astype(np.float32)


def test_oracle_recovers_cylinder_centres():
    cyls = [(2.0 * np.cos(a), 2.0 * np.sin(a), 0.3) for a in np.deg2rad([40.0, 130.0, 250.0])]
    st, r, b = so.scan(cylinder_scan(cyls, noise=0.002, seed=1))
    assert st == so.ST_OK and len(r) == 3
    assert np.allclose(r, 2.0, atol=0.03) and np.allclose(np.rad2deg(b), [40.0, 130.0, -110.0], atol=1.0)


def test_oracle_as_coded_quirks():
    # an obstacle dead ahead links beams 0 and 359: the reference asserts (sensor_landmark.cpp:69-81)
    st, r, b = so.scan(cylinder_scan([(2.0, 0.0, 0.3)]))
    assert st == so.ST_REF_ABORT and len(r) == 0
    # a cluster that runs up to beam 359 is never evaluated (the loop ends without a breaking beam, :92-121) ...
    c = 2.0 * np.cos(np.deg2rad(353.0)), 2.0 * np.sin(np.deg2rad(353.0))
    sc = cylinder_scan([(c[0], c[1], 0.25)])
    sc[0:3] = np.inf
    assert np.isfinite(sc[359]) and not np.isfinite(sc[0])
    st, r, b = so.scan(sc)
    assert st == so.ST_OK and len(r) == 0
    # ... and clusters of <= MIN_CLUSTER_POINTS points, or flat walls, give nothing
    st, r, b = so.scan(cylinder_scan([(6.0 * np.cos(1.0), 6.0 * np.sin(1.0), 0.12)]))
    assert len(r) == 0
    wall = np.full(360, np.inf, np.float32)
    wall[60:120] = (2.0 / np.cos(np.deg2rad(np.arange(60, 120) - 90.0))).astype(np.float32)   # the line y = 2
    st, r, b = so.scan(wall)
    assert st == so.ST_OK and len(r) == 0
    # empty scan
    st, r, b = so.scan(np.full(360, np.inf, np.float32))
    assert st == so.ST_OK and len(r) == 0


def test_oracle_beam_tables_are_libm_binary32():
    assert so.COS_MAP.dtype == np.float32 and so.COS_MAP[0] == 1.0 and so.SIN_MAP[0] == 0.0
    assert so.COS_MAP[90] != 0.0 and abs(float(so.COS_MAP[90])) < 1e-6      # cosf(float(pi/2)) is not zero
    assert np.allclose(so.SIN_MAP, np.sin(np.deg2rad(np.arange(360.0))), atol=3e-7)


@pytest.mark.gpu
def test_gpu_scan_parity(built):
    """Integer bookkeeping (status, landmark count per scan) bit-exact; range and bearing are binary32 results of a float
    pipeline with device acosf / double 4x4 algebra in place of glibc / LAPACK: within 1e-5 relative / 1e-5 rad."""
    from awesomeslam_amd.core import scan_landmarks

    scans = so.make_scans(600, seed=3)
    n, rg, bg, st = scan_landmarks(scans, max_out=32)
    tot = 0
    worst_r = worst_b = 0.0
    for i, s in enumerate(scans):
        ost, orr, ob = so.scan(s)
        assert st[i] == ost and n[i] == len(orr), i
        if len(orr):
            worst_r = max(worst_r, float(np.max(np.abs(rg[i, :n[i]] - orr) / orr)))
            worst_b = max(worst_b, float(np.max(np.abs(bg[i, :n[i]] - ob))))
            tot += len(orr)
    print(f"scan parity: {len(scans)} scans, {int((st == 1).sum())} reference aborts, {tot} landmarks, "
          f"max rel err range {worst_r:.2e}, max abs err bearing {worst_b:.2e} rad")
    assert tot > 1000 and worst_r < 1e-5 and worst_b < 1e-5


@pytest.mark.gpu
def test_gpu_scan_edge_cases(built):
    from awesomeslam_amd.core import scan_landmarks, SCAN_OVERFLOW, SCAN_REF_ABORT

    empty = np.full((1, 360), np.inf, np.float32)
    ahead = cylinder_scan([(2.0, 0.0, 0.3)])[None]
    many = cylinder_scan([(2.0 * np.cos(a), 2.0 * np.sin(a), 0.2) for a in np.deg2rad(np.arange(20.0, 340.0, 30.0))])[None]
    nan = np.full((1, 360), np.nan, np.float32)
    n, rg, bg, st = scan_landmarks(np.concatenate([empty, ahead, many, nan]), max_out=4)
    assert list(n[:2]) == [0, 0] and st[0] == 0 and st[1] == SCAN_REF_ABORT
    ost, orr, ob = so.scan(many[0])
    assert len(orr) > 4 and n[2] == 4 and st[2] == SCAN_OVERFLOW and np.allclose(rg[2], orr[:4], rtol=1e-5)
    assert n[3] == 0 and st[3] == 0
