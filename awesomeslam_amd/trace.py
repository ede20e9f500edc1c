"""Synthetic odom + range/bearing traces for the EKF/UKF-SLAM nodes, and their on-disk format.

A *trace* is what the reference's filter nodes see on their two topics, recorded in callback order
(reference: awesome_slam/src/ekf/ekf.cpp:74-114):

    /odom                  nav_msgs/Odometry          -> odom[t] = px, py, qw, qx, qy, qz, vx, wz
    /out/landmarks/sensor  awesome_slam_msgs/Landmarks -> obs[t][k] = (range, bearing)   (Landmarks.msg:1-2)

`obs_new[t] = 1` means a sensor message is delivered *before* odom message t (cbSensorLandmark replaces
the stored list, ekf.cpp:102-114); with 0 the node re-walks the list it already holds (ekf.cpp:147-150),
which is why the interleaving has to be part of the trace.  `dt[t]` is the binary32 `delta_time` the node
would have computed from ros::Time (ekf.cpp:80).

The generator is deterministic in (seed, trajectory index) and ROS-free.  Scenario: L
landmarks (pairwise spacing > 2 x MIN_DIST_THRESH) and a robot on a small closed loop.  Default layout
"field": the landmarks sit on a jittered grid EAST of the loop (nearest column FIELD_X0 = 12 m away), so
that every bearing stays well inside (-pi, pi): the reference UKF averages wrapped angles with a negative
central weight (ukf.cpp:296-303) and loses positive definiteness as soon as sigma-point bearings straddle
+-pi; layout "ring" (EKF only) puts them on four rings around the loop.  Two more properties of the
reference shape the scenario:
  * a landmark that is not re-observed keeps its stale range/bearing in Z and is still used in every
    update (ekf.cpp:175-181,300-310), so -- as in the reference's own Gazebo world, where all 8
    cylinders sit inside the lidar's range -- every mapped landmark must stay visible: the default
    sensor range is unlimited (`sensor_range` restricts it for divergence/edge tests only);
  * a landmark is promoted only after 10 sightings from (nearly) the same place, because the wait-list
    keeps sensor-frame readings and re-projects them from the current pose (ekf.cpp:217-253): the
    warm-up is therefore `stages` x 14 callbacks at rest at the start pose, and the landmarks are "born"
    (switched on) in `stages` groups, one per 14 callbacks, so that the state grows in several steps
    (ekf.cpp:255-290).
After the warm-up the robot drives on with arcs, exact straights (wz = 0), |wz| <= 0.001 segments, stops
and spins so that every branch of common.h:52-62 and ekf.cpp:206 is taken.

Ranges/bearings are stored as binary32: the wire type is float64 but the node narrows them at once
(LaserData::assign(const float &...), structures.h:96-100), so nothing is lost.
"""
from dataclasses import dataclass, field

import numpy as np

SENSOR_RANGE = float("inf")  # see module docstring; the reference world keeps every landmark in view
STOP_STEPS = 14  # > MIN_LANDMARK_OCC (config.h:44) callbacks at rest promote everything in view
RING_OFFSETS = (-2.25, -0.75, 0.75, 2.25)
TRACE_MAGIC = "aslam-trace-v1"
FILE_MAGIC = b"ASLTRC01"  # include/aslam_trace_file.h
NOISE_SLOTS = 48  # observation-noise draws per callback are max(NOISE_SLOTS, L): never a function of max_obs or B
OBS_CHUNK = 4096  # callbacks per vectorised chunk when building sensor messages


@dataclass
class Trace:
    """B trajectories x T callbacks (message level).  Indexing with an int gives one trajectory."""

    odom: np.ndarray  # [B, T, 8] float64
    dt: np.ndarray  # [B, T] float32
    obs_new: np.ndarray  # [B, T] uint8
    n_obs: np.ndarray  # [B, T] int32
    obs: np.ndarray  # [B, T, max_obs, 2] float32
    landmarks: np.ndarray  # [B, L, 2] float64 ground truth
    truth: np.ndarray  # [B, T, 3] float64 ground-truth pose after the motion of callback t
    warmup: int = 0  # callbacks in the survey lap (dimension may still grow before this)
    meta: dict = field(default_factory=dict)

    @property
    def B(self):
        return self.odom.shape[0]

    @property
    def T(self):
        return self.odom.shape[1]

    @property
    def L(self):
        return self.landmarks.shape[1]

    @property
    def max_obs(self):
        return self.obs.shape[2]

    def __getitem__(self, b):
        return Trajectory(self.odom[b], self.dt[b], self.obs_new[b], self.n_obs[b], self.obs[b],
                          self.landmarks[b], None if self.truth is None else self.truth[b], self.warmup)

    def select(self, ids):
        """The trajectories `ids` (in that order) as a trace of their own."""
        ids = list(ids)
        return Trace(self.odom[ids], self.dt[ids], self.obs_new[ids], self.n_obs[ids], self.obs[ids], self.landmarks[ids],
                     None if self.truth is None else self.truth[ids], self.warmup, dict(self.meta))

    def save(self, path):
        np.savez_compressed(path, magic=TRACE_MAGIC, odom=self.odom, dt=self.dt, obs_new=self.obs_new,
                            n_obs=self.n_obs, obs=self.obs, landmarks=self.landmarks, truth=self.truth,
                            warmup=self.warmup)

    # ---- the binary format of include/aslam_trace_file.h (what the C++ host library reads and writes)
    def to_file(self, path, with_truth=True):
        """Write an ASLTRC01 file: 64-byte header, then odom / dt / obs_new / n_obs / obs (+ ground truth), each array
        starting at a multiple of 64 bytes."""
        B, T = self.B, self.T
        has_truth = bool(with_truth and self.landmarks is not None and self.truth is not None and self.L > 0)
        hdr = np.zeros(64, np.uint8)
        hdr[0:8] = np.frombuffer(FILE_MAGIC, np.uint8)
        hdr[8:24] = np.array([B, T], "<i8").view(np.uint8)
        hdr[24:36] = np.array([self.max_obs, self.L if has_truth else 0, self.warmup], "<i4").view(np.uint8)
        arrays = [np.ascontiguousarray(self.odom, "<f8"), np.ascontiguousarray(self.dt, "<f4"),
                  np.ascontiguousarray(self.obs_new, np.uint8), np.ascontiguousarray(self.n_obs, "<i4"),
                  np.ascontiguousarray(self.obs, "<f4")]
        if has_truth:
            arrays += [np.ascontiguousarray(self.landmarks, "<f8"), np.ascontiguousarray(self.truth, "<f8")]
        with open(path, "wb") as f:
            f.write(hdr.tobytes())
            pos = 64
            for a in arrays:
                pad = (-pos) % 64
                f.write(b"\0" * pad)
                f.write(a.tobytes())
                pos += pad + a.nbytes

    @staticmethod
    def from_file(path):
        raw = np.fromfile(path, np.uint8)
        if raw.size < 64 or raw[0:8].tobytes() != FILE_MAGIC:
            raise ValueError(f"{path}: not an ASLTRC01 trace file")
        B, T = (int(v) for v in raw[8:24].view("<i8"))
        max_obs, L, warmup = (int(v) for v in raw[24:36].view("<i4"))
        pos = [64]

        def take(dtype, shape):
            pos[0] += (-pos[0]) % 64
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            if pos[0] + n > raw.size:
                raise ValueError(f"{path}: truncated")
            a = raw[pos[0]:pos[0] + n].view(dtype).reshape(shape).copy()
            pos[0] += n
            return a

        odom = take("<f8", (B, T, 8))
        dt = take("<f4", (B, T))
        obs_new = take(np.uint8, (B, T))
        n_obs = take("<i4", (B, T))
        obs = take("<f4", (B, T, max_obs, 2))
        landmarks = take("<f8", (B, L, 2)) if L > 0 else np.zeros((B, 0, 2))
        truth = take("<f8", (B, T, 3)) if L > 0 else None
        return Trace(odom, dt, obs_new, n_obs, obs, landmarks, truth, warmup)

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        if str(z["magic"]) != TRACE_MAGIC:
            raise ValueError(f"{path}: not an {TRACE_MAGIC} file")
        return Trace(z["odom"], z["dt"], z["obs_new"], z["n_obs"], z["obs"], z["landmarks"], z["truth"],
                     int(z["warmup"]))


@dataclass
class Trajectory:
    odom: np.ndarray  # [T, 8]
    dt: np.ndarray
    obs_new: np.ndarray
    n_obs: np.ndarray
    obs: np.ndarray  # [T, max_obs, 2] float32
    landmarks: np.ndarray
    truth: np.ndarray
    warmup: int = 0

    @property
    def T(self):
        return self.odom.shape[0]

    @property
    def max_obs(self):
        return self.obs.shape[1]

    def slice(self, t0, t1):
        """Callbacks t0 .. t1-1 as a trajectory of their own."""
        return Trajectory(self.odom[t0:t1], self.dt[t0:t1], self.obs_new[t0:t1], self.n_obs[t0:t1], self.obs[t0:t1],
                          self.landmarks, None if self.truth is None else self.truth[t0:t1], self.warmup)


def loop_radius(L):
    """Radius of the reference circle so that the inner ring holds L/4 landmarks >= 1.7 m apart."""
    per_ring = -(-L // 4)
    return max(4.0, per_ring * 1.7 / (2.0 * np.pi) + 2.25)


FIELD_X0 = 12.0  # nearest column of the "field" layout (m east of the loop centre)
FIELD_LOOP_RADIUS = 2.5


def _field_landmarks(L, rng):
    """Jittered 1.6 m grid east of the robot's loop: every landmark stays far (>= ~9 m) and strictly in
    the robot's x > 0 half-plane, so range/bearing are nearly linear over a 1 m prior and no sigma point
    straddles atan2's branch cut (the reference UKF averages unwrapped bearings, ukf.cpp:322-333)."""
    rows = int(np.ceil(np.sqrt(L * 1.5)))
    cols = -(-L // rows)
    gx, gy = np.meshgrid(np.arange(cols), np.arange(rows), indexing="ij")
    pts = np.stack([FIELD_X0 + 1.6 * gx.ravel(), 1.6 * (gy.ravel() - (rows - 1) / 2.0)], axis=1)[:L]
    pts = pts + rng.uniform(-0.25, 0.25, pts.shape)
    return pts[rng.permutation(L)]


def _landmarks(L, rng, layout="ring"):
    if layout == "field":
        return _field_landmarks(L, rng)
    Rc = loop_radius(L)
    pts = []
    for ring, off in enumerate(RING_OFFSETS):
        k = L // 4 + (1 if ring < L % 4 else 0)
        if k == 0:
            continue
        ang = (np.arange(k) + 0.25 * ring) * (2.0 * np.pi / k)
        r = Rc + off + rng.uniform(-0.1, 0.1, k)
        a = ang + rng.uniform(-0.1, 0.1, k) / (Rc + off)
        pts.append(np.stack([r * np.cos(a), r * np.sin(a)], axis=1))
    lm = np.concatenate(pts, axis=0)
    return lm[rng.permutation(L)]


def _schedule(T, L, rng, stages, warm_hop=0):
    """Pre-drawn control schedule: mode[t], v[t], wfix[t], warm-up length, birth time of each stage.
    mode: 0 arc (feedback), 1 exact straight, 2 tiny |wz| <= 0.001, 3 stop, 4 spin (wfix = turn rate)."""
    mode = np.zeros(T, np.int8)
    v = np.zeros(T)
    wfix = np.zeros(T)
    t = 0
    births = []
    for g in range(stages):
        # every warm-up stop is at the start pose: a wait-list entry is a sensor-frame reading that the
        # reference re-projects from the CURRENT pose (ekf.cpp:229-233), so a stop somewhere else would let
        # old entries capture the sightings of new landmarks
        # (warm_hop > 0 moves between the stops anyway: a stress case that fills the wait-list with junk)
        if g > 0 and warm_hop > 0:
            n = min(warm_hop, T - t)
            mode[t : t + n] = 0
            v[t : t + n] = 0.15
            t += n
        births.append(t)
        n = min(STOP_STEPS, T - t)
        mode[t : t + n] = 3
        t += n
    warm = t
    while t < T:
        u = rng.random()
        if u < 0.62:
            n, m, vv, ww = int(rng.integers(5, 40)), 0, float(rng.uniform(0.10, 0.22)), 0.0
        elif u < 0.74:
            n, m, vv, ww = int(rng.integers(2, 7)), 1, float(rng.uniform(0.10, 0.22)), 0.0
        elif u < 0.82:
            n, m, vv, ww = int(rng.integers(2, 6)), 2, float(rng.uniform(0.10, 0.22)), float(rng.choice([-1, 1]) * rng.uniform(2e-4, 1e-3))
        elif u < 0.92:
            n, m, vv, ww = int(rng.integers(1, 5)), 3, 0.0, 0.0
        else:
            n, m, vv, ww = 2 * int(rng.integers(1, 3)), 4, 0.0, float(rng.choice([-1, 1]) * rng.uniform(0.1, 0.4))
        n = min(n, T - t)
        mode[t : t + n] = m
        v[t : t + n] = vv
        if m == 4:  # turn one way, then back
            half = max(1, n // 2)
            wfix[t : t + half] = ww
            wfix[t + half : t + n] = -ww
        else:
            wfix[t : t + n] = ww
        t += n
    return mode, v, wfix, warm, births


def _wrap(a):
    return (a + np.pi) % (2.0 * np.pi) - np.pi


def make_traces(L, T, B=1, seed=0, dt_mode="fixed", sensor_every=1, stages=3, max_obs=None,
                odom_sigma=0.01, obs_sigma=0.02, first_traj=0, sensor_range=SENSOR_RANGE, layout="field",
                yaw_sigma=0.002, bearing_sigma=0.002, warm_hop=0):
    """Generate B trajectories (indices first_traj .. first_traj+B-1 of stream `seed`).

    layout "field" (default; both filters): landmarks on a grid east of a small loop.  layout "ring"
    (EKF only): four rings around the loop, the robot drives among them."""
    stages = max(1, min(stages, L))
    Rc = FIELD_LOOP_RADIUS if layout == "field" else loop_radius(L)
    rngs = [np.random.default_rng([seed, first_traj + b, L]) for b in range(B)]
    lms = np.stack([_landmarks(L, r, layout) for r in rngs])  # [B, L, 2]
    sched = [_schedule(T, L, r, stages, warm_hop) for r in rngs]
    mode = np.stack([s[0] for s in sched])
    vcmd = np.stack([s[1] for s in sched])
    wfix = np.stack([s[2] for s in sched])
    warm = max(s[3] for s in sched)
    # landmark j is switched on at the start of stop (j mod stages)
    birth = np.stack([np.asarray(s[4])[np.arange(L) % stages] for s in sched])  # [B, L]
    if dt_mode == "fixed":
        dt = np.ones((B, T), np.float32)
    elif dt_mode == "random":
        dt = np.stack([r.uniform(0.05, 1.0, T).astype(np.float32) for r in rngs])
    else:
        raise ValueError(dt_mode)

    # ---- ground truth: closed-loop unicycle (sequential in t, vectorised over B)
    truth = np.empty((B, T, 3))
    wcmd = np.empty((B, T))
    x = np.full(B, Rc)
    y = np.zeros(B)
    th = np.full(B, np.pi / 2)
    for t in range(T):
        m = mode[:, t]
        v = vcmd[:, t]
        rad = np.hypot(x, y)
        want = np.arctan2(y, x) + np.pi / 2 + np.clip(0.8 * (rad - Rc), -0.6, 0.6)
        w_fb = np.clip(v / Rc + 0.6 * _wrap(want - th), -0.5, 0.5)
        w = np.where(m == 0, w_fb, wfix[:, t])
        w = np.where((m == 0) & (np.abs(w) < 2e-3), 2e-3, w)  # keep feedback arcs on the arc branch
        h = dt[:, t].astype(np.float64)
        arc = np.abs(w) > 1e-3  # same split as common.h:52
        ws = np.where(arc, w, 1.0)
        x = x + np.where(arc, v / ws * (-np.sin(th) + np.sin(th + w * h)), v * h * np.cos(th))
        y = y + np.where(arc, v / ws * (np.cos(th) - np.cos(th + w * h)), v * h * np.sin(th))
        th = _wrap(th + w * h)
        truth[:, t, 0], truth[:, t, 1], truth[:, t, 2] = x, y, th
        wcmd[:, t] = w

    # ---- odometry messages: pose measured at the START of the motion the twist describes is what a real
    # odom stream gives; the reference uses pose as a direct measurement, so we publish the post-motion pose.
    odom = np.empty((B, T, 8))
    for b, r in enumerate(rngs):
        n = r.normal(0.0, 1.0, (T, 3)) * np.array([odom_sigma, odom_sigma, yaw_sigma])
        yaw = truth[b, :, 2] + n[:, 2]
        odom[b, :, 0] = truth[b, :, 0] + n[:, 0]
        odom[b, :, 1] = truth[b, :, 1] + n[:, 1]
        odom[b, :, 2] = np.cos(yaw / 2)
        odom[b, :, 3] = 0.0
        odom[b, :, 4] = 0.0
        odom[b, :, 5] = np.sin(yaw / 2)
    odom[:, :, 6] = vcmd
    odom[:, :, 7] = wcmd

    # ---- sensor messages (range, bearing) in scan order, binary32
    # (per trajectory and in fixed time chunks, so that trajectory b never depends on B)
    slots = max(NOISE_SLOTS, L) if max_obs is None else max_obs
    obs = np.zeros((B, T, slots, 2), np.float32)
    n_obs = np.zeros((B, T), np.int32)
    for b, r in enumerate(rngs):
        for t0 in range(0, T, OBS_CHUNK):
            t1 = min(T, t0 + OBS_CHUNK)
            dx = lms[b, None, :, 0] - truth[b, t0:t1, None, 0]
            dy = lms[b, None, :, 1] - truth[b, t0:t1, None, 1]
            rng_true = np.hypot(dx, dy)
            brg_true = np.arctan2(dy, dx) - truth[b, t0:t1, None, 2]  # deliberately not wrapped
            vis = (rng_true < sensor_range) & (np.arange(t0, t1)[:, None] >= birth[b][None, :])
            cnt = vis.sum(axis=1)
            if int(cnt.max()) > slots:
                raise ValueError(f"{int(cnt.max())} landmarks visible at once, only {slots} observation slots")
            key = np.where(vis, _wrap(brg_true), np.inf)
            order = np.argsort(key, axis=1, kind="stable")[:, :slots]
            rr = np.take_along_axis(rng_true, order, axis=1)
            bb = np.take_along_axis(brg_true, order, axis=1)
            noise = r.normal(0.0, 1.0, (t1 - t0, max(NOISE_SLOTS, L), 2))[:, : rr.shape[1]]
            noise = noise * np.array([obs_sigma, bearing_sigma])
            slot = np.arange(rr.shape[1])[None, :] < cnt[:, None]
            obs[b, t0:t1, : rr.shape[1], 0] = np.where(slot, rr + noise[:, :, 0], 0.0).astype(np.float32)
            obs[b, t0:t1, : rr.shape[1], 1] = np.where(slot, bb + noise[:, :, 1], 0.0).astype(np.float32)
            n_obs[b, t0:t1] = cnt
    if max_obs is None:  # trim to the widest message, rounded up to a multiple of 4
        max_obs = max(4, -(-int(n_obs.max()) // 4) * 4)
        obs = np.ascontiguousarray(obs[:, :, :max_obs])
    obs_new = np.zeros((B, T), np.uint8)
    obs_new[:, ::sensor_every] = 1
    if sensor_every > 1:
        # a re-walked message must be the one that was delivered: copy it forward so that the arrays
        # are self-describing (the nodes never look at obs[t] when obs_new[t] == 0)
        for t in range(T):
            if not obs_new[0, t]:
                obs[:, t] = obs[:, t - 1]
                n_obs[:, t] = n_obs[:, t - 1]
    return Trace(odom, dt, obs_new, n_obs, obs, lms, truth, warm,
                 {"L": L, "seed": seed, "dt_mode": dt_mode, "sensor_every": sensor_every, "first_traj": first_traj,
                  "stages": stages, "sensor_range": sensor_range, "layout": layout})


def full_dim(L):
    """State dimension once all L landmarks are promoted (ekf.cpp:261)."""
    return 3 + 2 * L


def dim_cap(L):
    """Smallest MAX_LANDMARK_COUNT (config.h:45; compared against the state DIMENSION, ekf.cpp:263) that
    admits L landmarks: growth is refused when N >= cap."""
    return full_dim(L) + 1
