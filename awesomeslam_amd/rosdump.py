"""Recorded ROS topics -> trace, without a ROS installation.

Input: the CSV text `rostopic echo -b run.bag -p /odom` and `rostopic echo -b run.bag -p /out/landmarks/sensor` print
(one row per message; the first row names the columns, `%time` is the receive stamp in ns).  Output: a message-level
`awesomeslam_amd.trace.Trace` for ONE robot, i.e. what the reference node's callbacks would have been handed:

* the node spins at FREQ = 1 Hz (`ekf.cpp:313-326`, `config.h`) with subscriber queues of size 1 (`ekf.cpp:41-42`): per spin
  it sees at most the LATEST message of each topic that arrived since the previous spin; older ones are dropped;
* a Landmarks message is delivered before the odometry message of the same spin (`obs_new = 1`); `LaserData::assign`
  narrows range/bearing to binary32 (`structures.h:85-101`);
* odometry callbacks before the first Landmarks message return early without touching anything (`ekf.cpp:76-77`): they
  are recorded (the trace replays that branch too), with `dt` as the node would compute it once it gets past the gate:
  `float delta_time = std::min(now - last_time, 1.0)` with `last_time` a BINARY32 member (`ekf.h:98`) that only callbacks past the gate
  advance (`ekf.cpp:79-81`) and that `initialize()` sets to `ros::Time::now().toSec()` when the node is constructed (`ekf.cpp:54`).  The
  narrowing matters: with epoch-sized stamps (1.7e9 s) a float has 128 s granularity, so the reference's `delta_time` is `now` minus a
  multiple of 128 s -- anything in (-64, 1.0] --, and with sim-time stamps of a few hundred seconds it still moves `dt` by up to 1.5e-5.
  This module reproduces that arithmetic exactly (np.float32 `last_time`).

A spin without a new odometry message runs no cbOdom and produces no callback in the trace (a Landmarks message that arrived
in it is kept for the next callback, as the node keeps `sensor_landmark`).
"""
import csv
import io

import numpy as np

from .trace import Trace

ODOM_FIELDS = ("field.pose.pose.position.x", "field.pose.pose.position.y", "field.pose.pose.orientation.w",
               "field.pose.pose.orientation.x", "field.pose.pose.orientation.y", "field.pose.pose.orientation.z",
               "field.twist.twist.linear.x", "field.twist.twist.angular.z")


def _rows(text):
    rd = csv.reader(io.StringIO(text))
    header = next(rd)
    return header, [r for r in rd if r]


def parse_odom_csv(text):
    """-> stamps [M] int64 ns, messages [M, 8] float64 (px, py, qw, qx, qy, qz, vx, wz)."""
    header, rows = _rows(text)
    col = {name: i for i, name in enumerate(header)}
    missing = [f for f in ("%time",) + ODOM_FIELDS if f not in col]
    if missing:
        raise ValueError(f"odometry dump lacks columns {missing}")
    t = np.array([int(r[col["%time"]]) for r in rows], np.int64)
    m = np.array([[float(r[col[f]]) for f in ODOM_FIELDS] for r in rows], np.float64).reshape(len(rows), 8)
    return t, m


def parse_landmarks_csv(text):
    """-> stamps [M] int64 ns, list of (range float64[k], bearing float64[k]).  awesome_slam_msgs/Landmarks is two
    float64 arrays of equal length (`Landmarks.msg:1-2`); a row is `%time, x0..x{k-1}, y0..y{k-1}`, ragged from row to row."""
    header, rows = _rows(text)
    if not header or header[0] != "%time":
        raise ValueError("landmark dump: first column must be %time")
    t, msgs = [], []
    for r in rows:
        vals = [v for v in r[1:] if v != ""]
        if len(vals) % 2:
            raise ValueError("landmark dump: x and y arrays of different length")
        k = len(vals) // 2
        t.append(int(r[0]))
        msgs.append((np.array(vals[:k], np.float64), np.array(vals[k:], np.float64)))
    return np.array(t, np.int64), msgs


def _to_sec(ns):
    """ros::Time::toSec(): (double)sec + 1e-9 * (double)nsec -- NOT ns * 1e-9, which rounds differently at epoch-sized stamps"""
    ns = int(ns)
    return float(ns // 10**9) + 1e-9 * float(ns % 10**9)


def to_trace(odom_csv, landmarks_csv, freq_hz=1.0, t_start_ns=None, t_init_ns=None):
    """Replay the node's spin loop over the two recorded topics.  `t_start_ns` is the time of the spin before the first one
    (the phase of the node's 1 Hz clock against the recording is not in the recording; default: the first message's stamp);
    `t_init_ns` is the time the node was constructed -- `initialize()` stores it in the binary32 `last_time` (`ekf.cpp:54`) -- and
    defaults to `t_start_ns`."""
    to, mo = parse_odom_csv(odom_csv)
    tl, ml = parse_landmarks_csv(landmarks_csv)
    if len(to) == 0:
        raise ValueError("no odometry messages")
    first = min(int(to[0]), int(tl[0]) if len(tl) else int(to[0]))
    t0 = first if t_start_ns is None else int(t_start_ns)
    period = int(round(1e9 / freq_hz))
    last_stamp = max(int(to[-1]), int(tl[-1]) if len(tl) else 0)
    odom, dts, new, nobs, obs = [], [], [], [], []
    io_, il = 0, 0
    pending = None        # a Landmarks message delivered in a spin without odometry
    gate_open = False     # init_z == false
    last_time = np.float32(_to_sec(t0 if t_init_ns is None else t_init_ns))  # ekf.cpp:54, narrowed by the float member (ekf.h:98)
    spin = t0
    while spin - period < last_stamp:
        spin += period    # messages with stamp <= spin are in the queues when spinOnce() runs
        lm = None
        while il < len(tl) and tl[il] <= spin:
            lm = ml[il]   # queue size 1: the latest wins
            il += 1
        od = None
        while io_ < len(to) and to[io_] <= spin:
            od = mo[io_]
            io_ += 1
        if lm is not None:
            pending = lm
        if od is None:
            continue
        now = _to_sec(spin)
        if pending is not None:
            gate_open = True
        dt = np.float32(min(now - float(last_time), 1.0))  # ekf.cpp:80: double - (double)float, min in double, narrowed to float
        if gate_open:
            last_time = np.float32(now)                   # ekf.cpp:81
        odom.append(od)
        dts.append(dt)
        if pending is not None:
            new.append(1)
            nobs.append(len(pending[0]))
            obs.append(np.stack([pending[0].astype(np.float32), pending[1].astype(np.float32)], -1))
            pending = None
        else:
            new.append(0)
            nobs.append(0)
            obs.append(np.zeros((0, 2), np.float32))
    T = len(odom)
    if T == 0:
        raise ValueError("no callbacks")
    max_obs = max(1, max(nobs))
    O = np.zeros((1, T, max_obs, 2), np.float32)
    for t, o in enumerate(obs):
        O[0, t, :len(o)] = o
    return Trace(np.array(odom, np.float64)[None], np.array(dts, np.float32)[None], np.array(new, np.uint8)[None],
                 np.array(nobs, np.int32)[None], O, np.zeros((1, 0, 2)), None, 0)


DUMP_T0_NS = 1_000_000_000_000  # spin phase dump_csv writes for (pass it to to_trace as t_start_ns)


def dump_csv(traj, freq_hz=1.0, t_start_ns=DUMP_T0_NS, extra_dropped=0, seed=0):
    """The inverse, for tests and examples: write a Trajectory as the two CSV texts.  Callback k is delivered in spin k+1;
    `extra_dropped` older messages per spin (with different content) exercise the size-1 queues."""
    rng = np.random.default_rng(seed)
    period = int(round(1e9 / freq_hz))
    oh = ["%time", "field.header.seq", "field.header.stamp", "field.header.frame_id", "field.child_frame_id"] + list(ODOM_FIELDS[:2]) + \
        ["field.pose.pose.position.z"] + [ODOM_FIELDS[3], ODOM_FIELDS[4], ODOM_FIELDS[5], ODOM_FIELDS[2]] + list(ODOM_FIELDS[6:])
    orow, lrow = [",".join(oh)], ["%time,field.x0,field.y0"]
    for k in range(traj.T):
        spin = t_start_ns + (k + 1) * period
        m = traj.odom[k]
        for e in range(extra_dropped, -1, -1):
            stamp = spin - period // 4 - e * (period // 8)
            v = m if e == 0 else m + rng.normal(size=8)
            vals = {f: repr(float(x)) for f, x in zip(ODOM_FIELDS, v)}
            vals.update({"%time": str(stamp), "field.header.seq": str(k), "field.header.stamp": str(stamp), "field.header.frame_id": "odom",
                         "field.child_frame_id": "base_footprint", "field.pose.pose.position.z": "0.0"})
            orow.append(",".join(vals[c] for c in oh))
        if traj.obs_new[k]:
            n = int(traj.n_obs[k])
            for e in range(extra_dropped, -1, -1):
                stamp = spin - period // 2 - e * (period // 8)
                o = traj.obs[k, :n].astype(np.float64) if e == 0 else rng.normal(size=(max(n - 1, 0), 2))
                lrow.append(",".join([str(stamp)] + [repr(float(x)) for x in o[:, 0]] + [repr(float(x)) for x in o[:, 1]]))
    return "\n".join(orow) + "\n", "\n".join(lrow) + "\n"
