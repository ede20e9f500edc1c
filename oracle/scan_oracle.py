"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's scan -> (range, bearing) front end
(`src/sensor_landmark/sensor_landmark.cpp:59-298`), as coded, binary32 where the reference computes in
`float`.  Never imported by the product path.  Parity unpinned: the reference has no tests and cannot be built here
(Eigen + ROS); the dense pieces (JacobiSVD, SelfAdjointEigenSolver, colPivHouseholderQr) are LAPACK's here.

As-coded behaviour that is kept (each cited below): the point that ends a cluster is not part of the next one; a cluster
that runs up to beam 359 is never evaluated; `std` divides by 5.0 whatever the cluster size; M is formed and not used.
If beams 0 and 359 are closer than MIN_DIST_THRESH the reference walks `theta` to 360 (after `p1 = p2` the loop condition
compares a point with itself) and fails `assert(theta < 359)` (`:81`; no NDEBUG in its CMakeLists): `scan()` reports that
as status REF_ABORT with no landmarks.
"""
import numpy as np

F = np.float32
MIN_DIST_THRESH = F(0.5)          # config.h:44
STD, MIN_MEAN, MAX_MEAN = F(0.4), F(1.5), F(3.0)   # config.h:49-51
MIN_CLUSTER_POINTS = 3            # config.h:52
DEG2RAD = F(0.01745329251)        # config.h:41
ST_OK, ST_REF_ABORT, ST_OVERFLOW = 0, 1, 2

# initialize(), sensor_landmark.cpp:49-56: std::sin / std::cos of a float argument are the binary32 overloads, i.e. the
# host libm's sinf / cosf (NumPy's float32 loops are its own SIMD kernels and differ in the last bit: glibc is asked)
import ctypes as _ct
_libm = _ct.CDLL("libm.so.6")
_libm.sinf.restype = _libm.cosf.restype = _ct.c_float
_libm.sinf.argtypes = _libm.cosf.argtypes = [_ct.c_float]
_TH = (DEG2RAD * np.arange(360, dtype=F)).astype(F)
SIN_MAP = np.array([_libm.sinf(float(t)) for t in _TH], F)
COS_MAP = np.array([_libm.cosf(float(t)) for t in _TH], F)


def dist(p, q):
    """Point::distance -> eulerDistance, structures.h:73-77, tools.h:53-59 (binary32 throughout)."""
    with np.errstate(invalid="ignore", over="ignore"):
        dx = F(np.float64(p[0]) - np.float64(q[0]))
        dy = F(np.float64(p[1]) - np.float64(q[1]))
        return F(np.sqrt(F(F(dx * dx) + F(dy * dy))))


def points_of(ranges):
    """bearing2pose, sensor_landmark.cpp:135-142"""
    r = np.asarray(ranges, F)
    with np.errstate(invalid="ignore", over="ignore"):
        return np.stack([(r * COS_MAP).astype(F), (r * SIN_MAP).astype(F)], -1)


def classify(pts):
    """circleClassification, sensor_landmark.cpp:147-186"""
    n = len(pts)
    P1, P2 = pts[0], pts[-1]
    angles = []
    s = F(0.0)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        for i in range(1, n - 1):
            P = pts[i]
            c, a, b = dist(P1, P2), dist(P1, P), dist(P2, P)
            num = F(F(F(a * a) + F(b * b)) - F(c * c))
            den = F(F(F(2.0) * a) * b)
            ang = F(np.arccos(F(num / den), dtype=F))
            angles.append(ang)
            s = F(s + ang)
        mean = F(s / F(n - 2))
        var = F(0.0)
        for v in angles:
            var = F(np.float64(var) + np.float64(F(v - mean)) ** 2)   # pow(float, int) -> double, accumulated into a float
        std = F(np.sqrt(np.float64(var) / 5.0))
    return bool(std < STD and mean > MIN_MEAN and mean < MAX_MEAN)


def fit(pts):
    """circleFitting, sensor_landmark.cpp:192-298 -> (centre x, centre y) as the binary32 values Circle is built from"""
    n = len(pts)
    xm = ym = F(0.0)
    for p in pts:
        xm = F(xm + p[0])
        ym = F(ym + p[1])
    xm, ym = F(xm / F(n)), F(ym / F(n))
    x = np.array([F(p[0] - xm) for p in pts], F)
    y = np.array([F(p[1] - ym) for p in pts], F)
    z = (x * x).astype(F) + (y * y).astype(F)
    zm = F(0.0)
    for v in z:
        zm = F(zm + v)
    zm = F(zm / F(n))
    Z = np.stack([z.astype(np.float64), x.astype(np.float64), y.astype(np.float64), np.ones(n)], -1)
    Hinv = np.array([[0, 0, 0, 0.5], [0, 1, 0, 0], [0, 0, 1, 0], [0.5, 0, 0, float(F(F(-2.0) * zm))]])
    U, sv, Vt = np.linalg.svd(Z, full_matrices=False)
    V = Vt.T
    if sv[3] > 10e-12:
        Y = V @ np.diag(sv) @ V.T
        Q = Y @ Hinv @ Y
        ev, evec = np.linalg.eigh((Q + Q.T) / 2)      # SelfAdjointEigenSolver reads one triangle
        sid, sev = 0, 99999.0
        for i in range(4):
            if ev[i] > 0 and ev[i] < sev:
                sid, sev = i, ev[i]
        A = np.linalg.solve(Y, evec[:, sid])
    else:
        A = V[:, 3]
    a = F(-A[1] / (2 * A[0]))
    b = F(-A[2] / (2 * A[0]))
    return F(a + xm), F(b + ym)


def to_laser(cx, cy):
    """Circle::toLaserData, structures.h:124-131 (double arithmetic on the stored floats, narrowed)"""
    cx, cy = np.float64(cx), np.float64(cy)
    return F(np.sqrt(cx * cx + cy * cy)), F(np.arctan2(cy, cx))


def scan(ranges, detail=False):
    """callback(), sensor_landmark.cpp:59-132 -> (status, range[], bearing[]) [+ the clusters that were fitted]"""
    p = points_of(ranges)
    if dist(p[0], p[359]) < MIN_DIST_THRESH:
        return (ST_REF_ABORT, np.zeros(0, F), np.zeros(0, F)) + (([],) if detail else ())
    out_r, out_b, fitted = [], [], []
    cluster = [0]
    p1 = 0
    for th in range(1, 360):
        if dist(p[p1], p[th]) < MIN_DIST_THRESH:
            cluster.append(th)
        elif len(cluster) > MIN_CLUSTER_POINTS and classify([p[i] for i in cluster]):
            cx, cy = fit([p[i] for i in cluster])    # (theta == 360 would join first_cluster: empty on this path)
            r, b = to_laser(cx, cy)
            out_r.append(r)
            out_b.append(b)
            fitted.append((cluster[0], cluster[-1]))
            cluster = []
        else:
            cluster = []
        p1 = th
    res = (ST_OK, np.array(out_r, F), np.array(out_b, F))
    return res + ((fitted,) if detail else ())


# ---- synthetic scans (test / bench input): cylinders around a robot, 1-degree beams, inf = no return
def make_scans(count, seed=0, n_cyl=(3, 10), radius=(0.12, 0.35), reach=(1.0, 6.0), noise=0.004, max_range=8.0):
    rng = np.random.default_rng(seed)
    out = np.full((count, 360), np.inf, F)
    th = np.deg2rad(np.arange(360.0))
    dx, dy = np.cos(th), np.sin(th)
    for s in range(count):
        k = int(rng.integers(n_cyl[0], n_cyl[1] + 1))
        ang = rng.uniform(0, 2 * np.pi, k)
        d = rng.uniform(reach[0], reach[1], k)
        rad = rng.uniform(radius[0], radius[1], k)
        cx, cy = d * np.cos(ang), d * np.sin(ang)
        best = np.full(360, np.inf)
        for j in range(k):
            bq = dx * cx[j] + dy * cy[j]
            disc = bq * bq - (cx[j] ** 2 + cy[j] ** 2 - rad[j] ** 2)
            hit = (disc > 0) & (bq - np.sqrt(np.maximum(disc, 0)) > 0.05)
            t = np.where(hit, bq - np.sqrt(np.maximum(disc, 0)), np.inf)
            best = np.minimum(best, t)
        best = np.where(best < max_range, best + rng.normal(0, noise, 360), np.inf)
        out[s] = best.astype(F)
    return out
