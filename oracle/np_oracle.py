"""np_oracle.py -- second, independent CPU restatement of the reference filter nodes (NumPy).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing under awesomeslam_amd/ does.

"parity unpinned" by the reference: iamarkaj/AwesomeSLAM has no tests, fixtures or golden vectors and
cannot be compiled in this image (no Eigen, no ROS).  This file and oracle/aslam_oracle.cpp were written
separately from the same reference text; tests/test_oracle_cross.py requires them to agree and
tests/golden/ freezes their outputs.

It restates (paths relative to /root/reference/awesome_slam):
    src/ekf/ekf.cpp:49-311, src/ukf/ukf.cpp:49-392, src/ukf/ukf.h:56-82,
    include/awesome_slam/common.h:46-90, tools.h:44-66, structures.h:44-112, config.h:39-65
and differs from the C++ oracle in how the dense algebra is carried out: products go through BLAS
(`@`), `MatrixXd::inverse()` through LAPACK getrf/getri (np.linalg.inv), `llt()` through LAPACK potrf
(np.linalg.cholesky).  Scalar libm calls (sin, cos, atan2, sqrt, fmod) go through Python's `math`
module / glibc so that every binary32 rounding point sees the same correctly-rounded inputs as the
reference would on this machine; `atan2f` is taken from libm via ctypes because NumPy's float32 loops
may use a vector library with different last-bit behaviour.
"""
import ctypes
import ctypes.util
import math

import numpy as np

f32 = np.float32
f64 = np.float64

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]

# ---- include/awesome_slam/config.h:39-65 (every constant is a binary32 value)
PI = f32(3.141592654)
MIN_DIST_THRESH = f32(0.5)
MIN_LANDMARK_OCC = 10
MAX_LANDMARK_COUNT = 30
UKF_STD_A = f32(0.2)
UKF_STD_YAW = f32(0.2)
UKF_KP_ROBOT_POSE = f32(0.001)
UKF_KP_LANDMARK_POSE = f32(1.0)
UKF_KR = f32(0.2)
UKF_KQ = f32(0.001)
EKF_KP_ROBOT_POSE = f32(0.001)
EKF_KR = f32(0.2)
EKF_KQ = f32(0.001)

TWO_PI = f32(f32(2) * PI)  # `2 * PI` is int * float -> float


# ---- tools.h
def normalize_angle(theta):
    """tools.h:44-50; the argument is converted to binary32 at the call, all arithmetic is binary32."""
    t = f32(theta)
    ret = f32(math.fmod(float(t), float(TWO_PI)))  # fmodf is exact, so fmod on the widened values is identical
    if ret > PI:
        ret = f32(ret - TWO_PI)
    if ret < -PI:
        ret = f32(ret + TWO_PI)
    return ret


def euler_distance(p1, p2):
    """tools.h:53-59; p1, p2 hold doubles (Eigen::Vector2d), differences and the norm are binary32."""
    dx = f32(f64(p1[0]) - f64(p2[0]))
    dy = f32(f64(p1[1]) - f64(p2[1]))
    return f32(np.sqrt(f32(f32(dx * dx) + f32(dy * dy))))


def quat2euler(w, x, y, z):
    """tools.h:62-66; the four arguments are converted to binary32."""
    w, x, y, z = f32(w), f32(x), f32(y), f32(z)
    num = f32(f32(2) * f32(f32(w * z) + f32(x * y)))
    den = f32(f32(1) - f32(f32(2) * f32(f32(z * z) + f32(y * y))))
    return f32(_libm.atan2f(float(num), float(den)))


# ---- structures.h
def to_point(rng, bearing, Z):
    """LaserData::toPoint, structures.h:104-111 -> the two binary32 coordinates (stored widened in a Vector2d)."""
    ang = float(Z[2]) + float(bearing)
    a = f32(float(Z[0]) + float(rng) * math.cos(ang))
    b = f32(float(Z[1]) + float(rng) * math.sin(ang))
    return (float(a), float(b))


# ---- common.h
def state_transition(N, point, vx, az, dt):
    """common.h:46-75.  vx, az, dt are binary32 (const float &)."""
    vx, az, dt = f32(vx), f32(az), f32(dt)
    P = np.array(point[:N], dtype=f64)
    th = float(point[2])
    azdt = float(f32(az * dt))
    if float(abs(az)) > 0.001:
        r = float(f32(vx / az))
        P[0] += r * (-math.sin(th) + math.sin(th + azdt))
        P[1] += r * (math.cos(th) - math.cos(th + azdt))
    else:
        vdt = float(f32(vx * dt))
        P[0] += vdt * math.cos(th)
        P[1] += vdt * math.sin(th)
    P[2] += azdt
    if len(point) > N:
        h = 0.5 * float(dt) * float(dt)
        P[0] += h * float(point[N]) * math.cos(th)
        P[1] += h * float(point[N]) * math.sin(th)
        P[2] += h * float(az)
    return P


def measurement(N, point):
    """common.h:78-90."""
    P = np.array(point, dtype=f64)
    for i in range(0, N - 3, 2):
        dx = float(point[3 + i]) - float(point[0])
        dy = float(point[4 + i]) - float(point[1])
        P[3 + i] = math.sqrt(dx * dx + dy * dy)
        P[4 + i] = math.atan2(dy, dx) - float(point[2])
    return P


def _resize_like(M, other):
    """Eigen conservativeResizeLike: other's shape and new coefficients, old block kept."""
    out = np.array(other, dtype=f64)
    sl = tuple(slice(0, min(a, b)) for a, b in zip(M.shape, out.shape))
    out[sl] = M[sl]
    return out


class NpFilter:
    """EKFSlam / UKFSlam without ROS.  kind: 'ekf' or 'ukf'."""

    def __init__(self, kind, max_landmark_count=MAX_LANDMARK_COUNT):
        assert kind in ("ekf", "ukf")
        self.kind = kind
        self.max_landmark_count = int(max_landmark_count)
        self.initialize()

    # ekf.cpp:49-71 / ukf.cpp:49-67
    def initialize(self):
        self.N = 3
        self.init_x = True
        self.init_z = True
        self.sensor = []  # list of [range f32, bearing f32]
        self.wait = []  # list of [range f32, bearing f32, count]
        N = self.N
        self.X = np.zeros(N)
        self.Z = np.zeros(N)
        self.Q = np.zeros((N, N))
        if self.kind == "ekf":
            self.A = np.eye(N)
            self.H = np.eye(N)
            self.I = np.eye(N)
            self.R = np.eye(N) * float(EKF_KR)
            self.P = np.eye(N) * float(EKF_KP_ROBOT_POSE)
            q = float(EKF_KQ)
        else:
            self.update_weights(3)
            self.R = np.eye(N) * float(UKF_KR)
            self.P = np.eye(N) * float(UKF_KP_ROBOT_POSE)
            q = float(UKF_KQ)
        self.Q[0, 0] = self.Q[1, 1] = self.Q[2, 2] = q

    # ukf.h:73-81
    def update_weights(self, N):
        self.lam = f32(3.0 - float(N + 2))
        den = f32(f32(self.lam + f32(N)) + f32(2))
        weight = f32(0.5 / float(den))
        self.weights = np.full(2 * N + 5, float(weight))
        self.weights[0] = float(f32(self.lam / den))

    # ekf.cpp:102-114
    def sensor_msg(self, xs, ys):
        self.init_z = False
        self.sensor = [[f32(a), f32(b)] for a, b in zip(xs, ys)]

    # ekf.cpp:217-253
    def _wait(self, rng, bearing):
        if not self.wait:
            self.wait.append([rng, bearing, 1])
            return
        p2 = to_point(rng, bearing, self.Z)
        best, mind = 0, None
        for i, (r, b, _) in enumerate(self.wait):
            d = euler_distance(p2, to_point(r, b, self.Z))
            if mind is None or d < mind:
                best, mind = i, d
        if mind < MIN_DIST_THRESH:
            self.wait[best][2] += 1
        else:
            self.wait.append([rng, bearing, 1])

    # ekf.cpp:255-290 / ukf.cpp:222-257
    def _grow(self, new):
        cacheN = self.N
        N = cacheN + 2 * len(new)
        if N >= self.max_landmark_count:
            return
        self.N = N
        self.X = _resize_like(self.X, np.zeros(N))
        self.Z = _resize_like(self.Z, np.zeros(N))
        self.Q = _resize_like(self.Q, np.zeros((N, N)))
        if self.kind == "ekf":
            self.I = _resize_like(self.I, np.eye(N))
            self.A = _resize_like(self.A, np.eye(N))
            self.H = _resize_like(self.H, np.eye(N))
        self.P = _resize_like(self.P, np.eye(N) * float(UKF_KP_LANDMARK_POSE))
        self.R = _resize_like(self.R, np.eye(N) * float(UKF_KR))
        if self.kind == "ukf":
            self.update_weights(N)
        for k, (rng, bearing) in enumerate(new):
            i = cacheN + 2 * k
            self.Z[i] = float(rng)
            self.Z[i + 1] = float(bearing)
            self.X[i] = self.Z[0] + self.Z[i] * math.cos(self.Z[2] + self.Z[i + 1])
            self.X[i + 1] = self.Z[1] + self.Z[i] * math.sin(self.Z[2] + self.Z[i + 1])

    # ekf.cpp:137-213 / ukf.cpp:113-180
    def _update_z(self, px, py, qw, qx, qy, qz, vx, wz, dt):
        self.Z[0] = px
        self.Z[1] = py
        self.Z[2] = float(quat2euler(qw, qx, qy, qz))
        for data in self.sensor:
            data[1] = normalize_angle(data[1])
            if self.N == 3:
                self._wait(data[0], data[1])
                continue
            p2 = to_point(data[0], data[1], self.Z)
            best, mind = 0, None
            for j in range(0, self.N - 3, 2):
                lm = (float(f32(self.X[3 + j])), float(f32(self.X[4 + j])))
                d = euler_distance(p2, lm)
                if mind is None or d < mind:
                    best, mind = j, d
            if mind < MIN_DIST_THRESH:
                self.Z[3 + best] = float(data[0])
                self.Z[4 + best] = float(data[1])
            else:
                self._wait(data[0], data[1])
        new = []
        for w in self.wait:
            if w[2] == MIN_LANDMARK_OCC:
                new.append((w[0], w[1]))
                w[2] += 1
        if new:
            self._grow(new)
        if self.kind == "ekf" and vx != 0.0 and wz != 0.0:
            dth = float(f32(float(wz) * float(f32(dt))))
            r = float(f32(float(vx) / float(wz)))
            z2 = float(self.Z[2])
            self.A[0, 0] = r * (-math.cos(z2) + math.cos(z2 + dth))
            self.A[1, 0] = r * (-math.sin(z2) + math.sin(z2 + dth))

    # ekf.cpp:117-134
    def _update_h(self):
        X, H = self.X, self.H
        for i in range(0, self.N - 3, 2):
            dx = float(X[3 + i]) - float(X[0])
            dy = float(X[4 + i]) - float(X[1])
            hyp = f32(dx * dx + dy * dy)
            dist = float(f32(np.sqrt(hyp)))
            hyp = float(hyp)
            H[3 + i, 0] = (-X[3 + i] + X[0]) / dist
            H[4 + i, 0] = -(-X[4 + i] + X[1]) / hyp
            H[3 + i, 1] = (-X[4 + i] + X[1]) / dist
            H[4 + i, 1] = (-X[3 + i] + X[0]) / hyp
            H[4 + i, 2] = -1.0
            H[3 + i, 3 + i] = -(-X[3 + i] + X[0]) / dist
            H[3 + i, 4 + i] = -(-X[4 + i] + X[1]) / dist
            H[4 + i, 3 + i] = (-X[4 + i] + X[1]) / hyp
            H[4 + i, 4 + i] = -(-X[3 + i] + X[0]) / hyp

    def _wrap_even(self, v):
        """normalizeAngle on entries 2, 4, ..., N-1 (ekf.cpp:304-307, ukf.cpp:336-339)."""
        for j in range(0, self.N - 1, 2):
            v[2 + j] = float(normalize_angle(v[2 + j]))

    # ekf.cpp:293-311
    def _slam_ekf(self, vx, az, dt):
        N = self.N
        self.X = state_transition(N, self.X, vx, az, dt)
        self.X[2] = float(normalize_angle(self.X[2]))
        self.P = self.A @ self.P @ self.A.T + self.Q
        self._update_h()
        S = self.H @ self.P @ self.H.T + self.R
        K = self.P @ self.H.T @ np.linalg.inv(S)
        Y = self.Z - measurement(N, self.X)
        self._wrap_even(Y)
        self.X = self.X + K @ Y
        self.P = (self.I - K @ self.H) @ self.P

    # ukf.cpp:260-392
    def _slam_ukf(self, vx, az, dt):
        N = self.N
        M = 2 * N + 5
        Xaug = np.zeros(N + 2)
        Xaug[:N] = self.X
        Paug = np.zeros((N + 2, N + 2))
        Paug[:N, :N] = self.P
        Paug[N, N] = float(f32(UKF_STD_A * UKF_STD_A))
        Paug[N + 1, N + 1] = float(f32(UKF_STD_YAW * UKF_STD_YAW))
        try:
            L = np.linalg.cholesky(Paug)
        except np.linalg.LinAlgError:  # Eigen's llt() does not report failure; propagate NaNs like it
            L = np.full_like(Paug, np.nan)
        w = float(f32(np.sqrt(f32(f32(self.lam + f32(N)) + f32(2)))))
        Xsig = np.empty((N + 2, M))
        Xsig[:, 0] = Xaug
        Xsig[:, 1 : N + 3] = Xaug[:, None] + w * L
        Xsig[:, N + 3 :] = Xaug[:, None] - w * L
        Xp = np.empty((N, M))
        for i in range(M):
            col = state_transition(N, Xsig[:, i], vx, az, dt)
            col[2] = float(normalize_angle(col[2]))
            Xp[:, i] = col
        W = self.weights
        self.X = Xp @ W
        D = Xp - self.X[:, None]
        for i in range(M):
            D[2, i] = float(normalize_angle(D[2, i]))
        self.P = (D * W) @ D.T + self.Q
        Zs = np.empty((N, M))
        for i in range(M):
            Zs[:, i] = measurement(N, Xp[:, i])
        Zpred = Zs @ W
        self._wrap_even(Zpred)
        DZ = Zs - Zpred[:, None]
        for i in range(M):
            col = DZ[:, i]
            self._wrap_even(col)
        S = (DZ * W) @ DZ.T + self.R
        Tc = (D * W) @ DZ.T
        K = Tc @ np.linalg.inv(S)
        Zdiff = self.Z - Zpred
        self._wrap_even(Zdiff)
        self.X = self.X + K @ Zdiff
        self.P = self.P - K @ S @ K.T

    def slam(self, vx, az, dt):
        if self.kind == "ekf":
            self._slam_ekf(vx, az, dt)
        else:
            self._slam_ukf(vx, az, dt)

    # ekf.cpp:74-99
    def odom_msg(self, px, py, qw, qx, qy, qz, vx, wz, dt):
        if self.init_z:
            return 0
        self._update_z(px, py, qw, qx, qy, qz, vx, wz, dt)
        if self.init_x:
            self.init_x = False
            self.X = self.Z.copy()
        self.slam(f32(vx), f32(wz), f32(dt))
        return 1

    def set_state(self, N, X, Z, P, a00=1.0, a10=0.0):
        """Synthetic state of dimension N for kernel-level tests (mirrors orc_set)."""
        self.initialize()
        self.init_z = False
        self.init_x = False
        self.N = N
        self.Q = _resize_like(self.Q, np.zeros((N, N)))
        if self.kind == "ekf":
            self.I = np.eye(N)
            self.A = np.eye(N)
            self.H = np.eye(N)
            self.A[0, 0] = a00
            self.A[1, 0] = a10
        self.R = np.eye(N) * float(UKF_KR)
        if self.kind == "ukf":
            self.update_weights(N)
        self.X = np.array(X, dtype=f64).copy()
        self.Z = np.array(Z, dtype=f64).copy()
        self.P = np.array(P, dtype=f64).reshape(N, N).copy()

    def replay(self, trace, T=None):
        """Run a message-level trace (awesomeslam_amd.trace.Trace, one trajectory); returns poses[T,3], dims[T]."""
        T = trace.T if T is None else T
        poses = np.zeros((T, 3))
        dims = np.zeros(T, dtype=np.int32)
        for t in range(T):
            if trace.obs_new[t]:
                k = int(trace.n_obs[t])
                self.sensor_msg(trace.obs[t, :k, 0], trace.obs[t, :k, 1])
            o = trace.odom[t]
            if self.odom_msg(o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], trace.dt[t]):
                poses[t] = self.X[:3]
            dims[t] = self.N
        return poses, dims
