"""ctypes binding of libaslam_core.so / libaslam_node.so (include/aslam_core.h, csrc/host/aslam_node.h).

This is plumbing: every number comes out of the HIP kernels.  There is no CPU fallback -- if the shared
library is missing, or no HIP device is present, the calls raise (AslamError / OSError).
"""
import ctypes
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_CORE = os.path.join(_CSRC, "libaslam_core.so")
_NODE = os.path.join(_CSRC, "libaslam_node.so")

EKF, UKF = 0, 1
F64, F32 = 0, 1
FILTERS = {"ekf": EKF, "ukf": UKF}

ST_GROWTH_REFUSED, ST_WAIT_OVERFLOW, ST_NOT_PD, ST_OBS_OVERFLOW, ST_INTERNAL = 1, 2, 4, 8, 16

# every symbol include/aslam_core.h declares (tests check the library exports them all)
CORE_SYMBOLS = (
    "aslam_create", "aslam_destroy", "aslam_reset", "aslam_last_error", "aslam_abi_version", "aslam_set_state",
    "aslam_grow", "aslam_ekf_step", "aslam_ukf_step", "aslam_ekf_step_batch", "aslam_ukf_step_batch", "aslam_set_trace", "aslam_replay", "aslam_get_dim",
    "aslam_get_state", "aslam_get_A", "aslam_get_landmarks", "aslam_get_wait", "aslam_get_status",
    "aslam_get_layout", "aslam_kernel_info", "aslam_get_launch_info",
)
NODE_SYMBOLS = (
    "aslam_node_create", "aslam_node_create_at", "aslam_node_destroy", "aslam_node_error", "aslam_node_sensor", "aslam_node_odom",
    "aslam_node_odom_now", "aslam_node_dim", "aslam_node_get", "aslam_node_wait", "aslam_node_core",
    "aslam_host_narrow_odom",
)
TRACE_FILE_SYMBOLS = (
    "aslam_trace_file_open", "aslam_trace_file_close", "aslam_trace_file_error", "aslam_trace_file_dims",
    "aslam_trace_file_view", "aslam_trace_file_raw", "aslam_trace_file_write",
)


class AslamError(RuntimeError):
    pass


class Config(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in
                ("filter", "dtype", "max_landmark_count", "batch", "max_obs", "max_wait", "device", "reserved")]


class TraceView(ctypes.Structure):
    _fields_ = [("T", ctypes.c_int64), ("max_obs", ctypes.c_int32), ("is_device", ctypes.c_int32),
                ("pose", ctypes.c_void_p), ("yaw", ctypes.c_void_p), ("twist", ctypes.c_void_p),
                ("dt", ctypes.c_void_p), ("obs_new", ctypes.c_void_p), ("n_obs", ctypes.c_void_p),
                ("obs", ctypes.c_void_p)]


_BUILT_HERE = False  # set by build() when THIS process's make run (re)compiled libaslam_core.so


def _sha256(path):
    import hashlib

    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def build(force=False, ukf=True):
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU): `make all`, which decides itself what is stale.  The Makefile scans
    the device assembly of the very compilation that produces libaslam_core.so (tools/check_spill_exec.py, check_agpr_strip.py,
    check_vmcnt_protocol.py), moves the library into place only when all guards are clean -- a library that failed one is never left where
    core_lib() would load it; GUARDS=0 installs under another name -- and THEN writes csrc/build_info.json (sha256 of the library, hash of
    the sources, guards, compiler, time) from the same rule, so the record always describes the file under the loader's name."""
    global _BUILT_HERE
    have_ukf = ukf and os.path.exists(os.path.join(_CSRC, "ukf_small.h"))
    before = _sha256(_CORE) if os.path.exists(_CORE) else None
    cmd = ["make", "-s", "-C", _CSRC, f"UKF={1 if have_ukf else 0}", "GUARDS=1"] + (["-B"] if force else []) + ["all"]
    subprocess.check_call(cmd)
    _BUILT_HERE = _BUILT_HERE or before != _sha256(_CORE)
    return _CORE, _NODE


def build_info():
    """The record the Makefile wrote for csrc/libaslam_core.so, CHECKED against the file that is there now: "matches_library" is True only
    when the sha256 in the record is the sha256 of the library core_lib() loads (False: the record describes some other binary -- the library
    was replaced behind the Makefile's back; None: no record).  "reused_here": this process did not compile it (it came with the snapshot)."""
    import json

    p = os.path.join(_CSRC, "build_info.json")
    try:
        info = json.load(open(p))
    except (OSError, ValueError):
        info = {"built": "unknown (no build_info.json: the library was not installed by csrc/Makefile)"}
    info.pop("sources", None)
    if "sha256" in info and os.path.exists(_CORE):
        info["matches_library"] = info["sha256"] == _sha256(_CORE)
        if not info["matches_library"]:
            info["built"] = "unknown (build_info.json does not describe the libaslam_core.so that is loaded)"
    else:
        info["matches_library"] = None
    info["reused_here"] = not _BUILT_HERE
    return info


_core = None
_node = None


def core_lib():
    global _core
    if _core is None:
        if not os.path.exists(_CORE):
            raise OSError(f"{_CORE} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the filter core)")
        # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's libamdhip64.so.7): it has to be in
        # the process first so that libaslam_core.so binds to that one copy instead of loading a second runtime
        import torch  # noqa: F401

        L = ctypes.CDLL(_CORE)
        vp, ci, cf, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double
        pd, pf, pi, pu = (ctypes.POINTER(t) for t in (ctypes.c_double, ctypes.c_float, ctypes.c_int, ctypes.c_uint32))
        L.aslam_last_error.restype = ctypes.c_char_p
        L.aslam_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
        L.aslam_destroy.argtypes = [vp]
        L.aslam_reset.argtypes = [vp]
        L.aslam_set_state.argtypes = [vp, ci, ci, pd, pd, pd]
        L.aslam_grow.argtypes = [vp, ci, ci, pd, pd]
        L.aslam_ekf_step.argtypes = [vp, ci, cf, cf, cf, pd, cd, cd, pd, vp]
        L.aslam_ukf_step.argtypes = [vp, ci, cf, cf, cf, pd, pd, vp]
        L.aslam_ekf_step_batch.argtypes = [vp, pf, pf, pf, pd, ci, pd, pd, pd, ci, vp]
        L.aslam_ukf_step_batch.argtypes = [vp, pf, pf, pf, pd, ci, pd, ci, vp]
        L.aslam_set_trace.argtypes = [vp, ctypes.POINTER(TraceView)]
        L.aslam_replay.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, vp, vp]
        L.aslam_get_dim.argtypes = [vp, ci, pi]
        L.aslam_get_state.argtypes = [vp, ci, pd, pd, pd]
        L.aslam_get_A.argtypes = [vp, ci, pd, pd]
        L.aslam_get_landmarks.argtypes = [vp, ci, pd, pd, pi]
        L.aslam_get_wait.argtypes = [vp, ci, pf, pf, pu, ci, pi]
        L.aslam_get_status.argtypes = [vp, ci, pu]
        L.aslam_get_layout.argtypes = [vp, pi, ctypes.POINTER(ctypes.c_int64)]
        L.aslam_kernel_info.argtypes = [vp, ctypes.c_char_p, ci, pi, pi, pi]
        L.aslam_get_launch_info.argtypes = [vp, pi, pi, pi]
        _core = L
    return _core


def node_lib():
    global _node
    if _node is None:
        if not os.path.exists(_NODE):
            raise OSError(f"{_NODE} is missing: run __graft_entry__.build()")
        core_lib()
        L = ctypes.CDLL(_NODE)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        pd, pf, pu = (ctypes.POINTER(t) for t in (ctypes.c_double, ctypes.c_float, ctypes.c_uint32))
        L.aslam_node_create.restype = vp
        L.aslam_node_create.argtypes = [ci, ci, ci]
        L.aslam_node_create_at.restype = vp
        L.aslam_node_create_at.argtypes = [ci, ci, ci, ctypes.c_double]
        L.aslam_node_destroy.argtypes = [vp]
        L.aslam_node_error.restype = ctypes.c_char_p
        L.aslam_node_sensor.argtypes = [vp, ci, pd, pd]
        L.aslam_node_odom.argtypes = [vp, pd, ctypes.c_float]
        L.aslam_node_odom_now.argtypes = [vp, pd, ctypes.c_double]
        L.aslam_node_dim.argtypes = [vp]
        L.aslam_node_get.argtypes = [vp, pd, pd, pd, pd]
        L.aslam_node_wait.argtypes = [vp, pf, pf, pu, ci]
        L.aslam_node_core.restype = vp
        L.aslam_node_core.argtypes = [vp]
        L.aslam_host_narrow_odom.argtypes = [ctypes.c_int64, pd, pd, pf, pd]
        # include/aslam_trace_file.h
        L.aslam_trace_file_error.restype = ctypes.c_char_p
        L.aslam_trace_file_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        L.aslam_trace_file_close.argtypes = [vp]
        L.aslam_trace_file_close.restype = None
        L.aslam_trace_file_dims.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        L.aslam_trace_file_view.argtypes = [vp, ctypes.POINTER(TraceView)]
        L.aslam_trace_file_raw.argtypes = [vp] + [ctypes.POINTER(vp)] * 5
        L.aslam_trace_file_write.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                             vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp]
        _node = L
    return _node


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _chk(rc):
    if rc != 0:
        raise AslamError(f"aslam_core error {rc}: {core_lib().aslam_last_error().decode()}")


SCAN_SYMBOLS = ("aslam_scan_landmarks",)
SCAN_BEAMS = 360
SCAN_REF_ABORT, SCAN_OVERFLOW = 1, 2


def scan_landmarks(ranges, max_out=64, device=0):
    """include/aslam_scan.h on host arrays: ranges [count, 360] f32 -> (n [count] i32, range [count, max_out] f32,
    bearing [count, max_out] f32, status [count] u32).  Runs on the GPU; there is no CPU fallback."""
    r = np.ascontiguousarray(ranges, np.float32)
    if r.ndim != 2 or r.shape[1] != SCAN_BEAMS:
        raise ValueError("ranges must be [count, 360]")
    cnt = r.shape[0]
    rg = np.zeros((cnt, max_out), np.float32)
    bg = np.zeros((cnt, max_out), np.float32)
    n = np.zeros(cnt, np.int32)
    st = np.zeros(cnt, np.uint32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib = core_lib()
    lib.aslam_scan_landmarks.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    _chk(lib.aslam_scan_landmarks(vp(r), cnt, 0, int(max_out), vp(rg), vp(bg), vp(n), vp(st), int(device), None))
    return n, rg, bg, st


def scan_landmarks_device(ranges_ptr, count, max_out, range_ptr, bearing_ptr, n_ptr, status_ptr, device=0, stream=None):
    """include/aslam_scan.h on device pointers (ints), asynchronous on `stream`."""
    lib = core_lib()
    lib.aslam_scan_landmarks.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    _chk(lib.aslam_scan_landmarks(ranges_ptr, int(count), 1, int(max_out), range_ptr, bearing_ptr, n_ptr, status_ptr, int(device), stream))


class TraceFile:
    """A trace file opened by the C++ host library (include/aslam_trace_file.h)."""

    def __init__(self, path):
        h = ctypes.c_void_p()
        if node_lib().aslam_trace_file_open(os.fsencode(path), ctypes.byref(h)) != 0:
            raise AslamError(node_lib().aslam_trace_file_error().decode())
        self._h = h
        B, T, mo, L = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        node_lib().aslam_trace_file_dims(h, ctypes.byref(B), ctypes.byref(T), ctypes.byref(mo), ctypes.byref(L))
        self.B, self.T, self.max_obs, self.L = B.value, T.value, mo.value, L.value

    def view(self):
        """The narrowed aslam_trace view (host pointers owned by this object)."""
        v = TraceView()
        if node_lib().aslam_trace_file_view(self._h, ctypes.byref(v)) != 0:
            raise AslamError(node_lib().aslam_trace_file_error().decode())
        return v

    def arrays(self):
        """NumPy copies of the narrowed view: pose, yaw, twist, dt, obs_new, n_obs, obs."""
        v = self.view()
        B, T, mo = self.B, self.T, self.max_obs

        def arr(ptr, ct, shape):
            n = int(np.prod(shape))
            if n == 0:
                return np.zeros(shape, ct)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), (n,)).reshape(shape).copy()

        return (arr(v.pose, ctypes.c_double, (B, T, 2)), arr(v.yaw, ctypes.c_float, (B, T)),
                arr(v.twist, ctypes.c_double, (B, T, 2)), arr(v.dt, ctypes.c_float, (B, T)),
                arr(v.obs_new, ctypes.c_uint8, (B, T)), arr(v.n_obs, ctypes.c_int32, (B, T)),
                arr(v.obs, ctypes.c_float, (B, T, mo, 2)))

    def close(self):
        if getattr(self, "_h", None):
            try:
                node_lib().aslam_trace_file_close(self._h)
            except TypeError:
                pass
            self._h = None

    __del__ = close

    @staticmethod
    def write(path, tr, with_truth=True):
        """Write a trace.Trace through the C++ writer."""
        p = lambda a, dt: np.ascontiguousarray(a, dt)  # noqa: E731
        odom, dt, new, nobs, obs = p(tr.odom, np.float64), p(tr.dt, np.float32), p(tr.obs_new, np.uint8), \
            p(tr.n_obs, np.int32), p(tr.obs, np.float32)
        has = bool(with_truth and tr.truth is not None and tr.L > 0)
        lm = p(tr.landmarks, np.float64) if has else None
        truth = p(tr.truth, np.float64) if has else None
        vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        rc = node_lib().aslam_trace_file_write(os.fsencode(path), tr.B, tr.T, tr.max_obs, int(tr.warmup), vp(odom), vp(dt),
                                               vp(new), vp(nobs), vp(obs), tr.L if has else 0, vp(lm), vp(truth))
        if rc != 0:
            raise AslamError(node_lib().aslam_trace_file_error().decode())


def narrow_odom(odom):
    """Message-level odometry [..., 8] -> (pose [..., 2] f64, yaw [...] f32, twist [..., 2] f64), done by the
    C++ host mirror so that quat2euler is the host libm's atan2f (tools.h:62-66)."""
    odom = np.ascontiguousarray(odom, np.float64)
    lead = odom.shape[:-1]
    cnt = int(np.prod(lead))
    pose = np.empty(lead + (2,), np.float64)
    yaw = np.empty(lead, np.float32)
    twist = np.empty(lead + (2,), np.float64)
    node_lib().aslam_host_narrow_odom(cnt, _ptr(odom, ctypes.c_double), _ptr(pose, ctypes.c_double),
                                      _ptr(yaw, ctypes.c_float), _ptr(twist, ctypes.c_double))
    return pose, yaw, twist


class Core:
    """One libaslam_core context: `batch` independent filters resident on one GPU."""

    def __init__(self, filter="ekf", max_landmark_count=30, batch=1, max_obs=16, max_wait=128, device=0, dtype=F64):
        self.filter = filter
        self.batch = int(batch)
        self.max_obs = int(max_obs)
        cfg = Config(FILTERS[filter], dtype, int(max_landmark_count), int(batch), int(max_obs), int(max_wait),
                     int(device), 0)
        h = ctypes.c_void_p()
        _chk(core_lib().aslam_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self._trace_keep = None
        self.T = 0

    def close(self):
        if getattr(self, "_h", None):
            try:
                core_lib().aslam_destroy(self._h)
            except TypeError:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    __del__ = close

    def reset(self):
        _chk(core_lib().aslam_reset(self._h))

    # ---- per-callback seam
    def set_state(self, traj, n, X=None, Z=None, P=None):
        X, Z, P = (None if a is None else np.ascontiguousarray(a, np.float64) for a in (X, Z, P))
        _chk(core_lib().aslam_set_state(self._h, traj, int(n), _ptr(X, ctypes.c_double), _ptr(Z, ctypes.c_double),
                                        _ptr(P, ctypes.c_double)))

    def grow(self, traj, n_new, x_seed, z_seed):
        x_seed = np.ascontiguousarray(x_seed, np.float64)
        z_seed = np.ascontiguousarray(z_seed, np.float64)
        _chk(core_lib().aslam_grow(self._h, traj, int(n_new), _ptr(x_seed, ctypes.c_double), _ptr(z_seed, ctypes.c_double)))

    def ekf_step(self, traj, vx, az, dt, Z, a00, a10, stream=None):
        Z = np.ascontiguousarray(Z, np.float64)
        X = np.empty(len(Z))
        _chk(core_lib().aslam_ekf_step(self._h, traj, float(np.float32(vx)), float(np.float32(az)), float(np.float32(dt)),
                                       _ptr(Z, ctypes.c_double), float(a00), float(a10), _ptr(X, ctypes.c_double), stream))
        return X

    def ukf_step(self, traj, vx, az, dt, Z, stream=None):
        Z = np.ascontiguousarray(Z, np.float64)
        X = np.empty(len(Z))
        _chk(core_lib().aslam_ukf_step(self._h, traj, float(np.float32(vx)), float(np.float32(az)), float(np.float32(dt)),
                                       _ptr(Z, ctypes.c_double), _ptr(X, ctypes.c_double), stream))
        return X

    def step_batch(self, vx, az, dt, Z, a00=None, a10=None, X_out=None, stream=None):
        """slam() for all filters of the context in one launch chain.  vx, az, dt: [batch] float32; Z: [batch, ldz] float64;
        a00, a10: [batch] float64 (EKF).  Asynchronous: the arrays are used until `stream` is synchronised; X_out ([batch, ldx]
        float64, optional) is valid only after that.  Arrays are taken as they are (no copies): pass C-contiguous ones."""
        for a, t in ((vx, np.float32), (az, np.float32), (dt, np.float32), (Z, np.float64)):
            assert a.dtype == t and a.flags.c_contiguous and a.shape[0] == self.batch
        ldx = 0 if X_out is None else X_out.shape[1]
        if self.filter == "ekf":
            assert a00.dtype == np.float64 and a10.dtype == np.float64
            _chk(core_lib().aslam_ekf_step_batch(self._h, _ptr(vx, ctypes.c_float), _ptr(az, ctypes.c_float), _ptr(dt, ctypes.c_float),
                                                 _ptr(Z, ctypes.c_double), Z.shape[1], _ptr(a00, ctypes.c_double), _ptr(a10, ctypes.c_double),
                                                 _ptr(X_out, ctypes.c_double), ldx, stream))
        else:
            _chk(core_lib().aslam_ukf_step_batch(self._h, _ptr(vx, ctypes.c_float), _ptr(az, ctypes.c_float), _ptr(dt, ctypes.c_float),
                                                 _ptr(Z, ctypes.c_double), Z.shape[1], _ptr(X_out, ctypes.c_double), ldx, stream))

    def sync(self, stream=None):
        import torch
        torch.cuda.synchronize() if stream is None else torch.cuda.ExternalStream(stream).synchronize()

    # ---- replay seam
    def set_trace(self, trace):
        """Bind a message-level awesomeslam_amd.trace.Trace (B == batch): narrowed on the host, copied to HBM."""
        if trace.B != self.batch or trace.max_obs != self.max_obs:
            raise ValueError(f"trace is B={trace.B}, max_obs={trace.max_obs}; context is B={self.batch}, max_obs={self.max_obs}")
        pose, yaw, twist = narrow_odom(trace.odom)
        arrs = dict(pose=pose, yaw=yaw, twist=twist, dt=np.ascontiguousarray(trace.dt, np.float32),
                    obs_new=np.ascontiguousarray(trace.obs_new, np.uint8),
                    n_obs=np.ascontiguousarray(trace.n_obs, np.int32), obs=np.ascontiguousarray(trace.obs, np.float32))
        tv = TraceView(trace.T, trace.max_obs, 0, *(arrs[k].ctypes.data for k in ("pose", "yaw", "twist", "dt", "obs_new", "n_obs", "obs")))
        _chk(core_lib().aslam_set_trace(self._h, ctypes.byref(tv)))
        self.T = trace.T

    def set_trace_file(self, tf):
        """Bind a TraceFile: the C++ host library narrows the messages, aslam_set_trace copies the view to HBM."""
        if tf.B != self.batch or tf.max_obs != self.max_obs:
            raise ValueError(f"trace file is B={tf.B}, max_obs={tf.max_obs}; context is B={self.batch}, max_obs={self.max_obs}")
        v = tf.view()
        _chk(core_lib().aslam_set_trace(self._h, ctypes.byref(v)))
        self.T = tf.T

    def set_trace_device(self, T, pose, yaw, twist, dt, obs_new, n_obs, obs):
        """Bind device-resident arrays (raw device pointers as ints, e.g. torch.Tensor.data_ptr())."""
        tv = TraceView(int(T), self.max_obs, 1, pose, yaw, twist, dt, obs_new, n_obs, obs)
        _chk(core_lib().aslam_set_trace(self._h, ctypes.byref(tv)))
        self.T = int(T)

    def replay(self, t0, nsteps, poses_ptr=None, dims_ptr=None, stream=None):
        """Asynchronous.  poses_ptr / dims_ptr: raw device pointers ([batch][nsteps][3] f64, [batch][nsteps] i32)."""
        _chk(core_lib().aslam_replay(self._h, int(t0), int(nsteps), poses_ptr, dims_ptr, stream))

    # ---- read-back
    def dim(self, traj=0):
        n = ctypes.c_int()
        _chk(core_lib().aslam_get_dim(self._h, traj, ctypes.byref(n)))
        return n.value

    def state(self, traj=0, with_P=True):
        n = self.dim(traj)
        X, Z = np.empty(n), np.empty(n)
        P = np.empty((n, n)) if with_P else None
        _chk(core_lib().aslam_get_state(self._h, traj, _ptr(X, ctypes.c_double), _ptr(Z, ctypes.c_double), _ptr(P, ctypes.c_double)))
        return X, Z, P

    def A(self, traj=0):
        a, b = ctypes.c_double(), ctypes.c_double()
        _chk(core_lib().aslam_get_A(self._h, traj, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def landmarks(self, traj=0):
        n = self.dim(traj)
        L = (n - 3) // 2
        x, y = np.empty(L), np.empty(L)
        k = ctypes.c_int()
        _chk(core_lib().aslam_get_landmarks(self._h, traj, _ptr(x, ctypes.c_double), _ptr(y, ctypes.c_double), ctypes.byref(k)))
        return x[: k.value], y[: k.value]

    def wait_list(self, traj=0, cap=512):
        r, b, c = np.empty(cap, np.float32), np.empty(cap, np.float32), np.empty(cap, np.uint32)
        k = ctypes.c_int()
        _chk(core_lib().aslam_get_wait(self._h, traj, _ptr(r, ctypes.c_float), _ptr(b, ctypes.c_float), _ptr(c, ctypes.c_uint32), cap, ctypes.byref(k)))
        k = min(k.value, cap)
        return r[:k], b[:k], c[:k]

    def status(self, traj=0):
        s = ctypes.c_uint32()
        _chk(core_lib().aslam_get_status(self._h, traj, ctypes.byref(s)))
        return s.value

    def layout(self):
        npad, nbytes = ctypes.c_int(), ctypes.c_int64()
        _chk(core_lib().aslam_get_layout(self._h, ctypes.byref(npad), ctypes.byref(nbytes)))
        return npad.value, nbytes.value

    def kernel_info(self):
        name = ctypes.create_string_buffer(192)
        g, b, l = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _chk(core_lib().aslam_kernel_info(self._h, name, 192, ctypes.byref(g), ctypes.byref(b), ctypes.byref(l)))
        return {"name": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}

    def launch_info(self):
        """What the last replay / step of this context really launched (aslam_get_launch_info)."""
        g, r, l = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _chk(core_lib().aslam_get_launch_info(self._h, ctypes.byref(g), ctypes.byref(r), ctypes.byref(l)))
        return {"stream_groups": g.value, "chol_resident": bool(r.value), "launches_per_callback": l.value}


class Node:
    """aslam::EKFSlam / aslam::UKFSlam host mirror (C++), driving the per-callback seam of the core."""

    def __init__(self, filter="ekf", max_landmark_count=30, device=0, now_init=0.0):
        """`now_init`: the clock reading (seconds) at construction -- initialize()'s last_time = ros::Time::now().toSec() (ekf.cpp:54)"""
        h = node_lib().aslam_node_create_at(FILTERS[filter], int(max_landmark_count), int(device), float(now_init))
        if not h:
            raise AslamError(node_lib().aslam_node_error().decode())
        self._h = ctypes.c_void_p(h)

    def close(self):
        if getattr(self, "_h", None):
            node_lib().aslam_node_destroy(self._h)
            self._h = None

    __del__ = close

    def sensor_msg(self, xs, ys):
        xs = np.ascontiguousarray(xs, np.float64)
        ys = np.ascontiguousarray(ys, np.float64)
        node_lib().aslam_node_sensor(self._h, len(xs), _ptr(xs, ctypes.c_double), _ptr(ys, ctypes.c_double))

    def odom_msg(self, msg8, dt):
        m = np.ascontiguousarray(msg8, np.float64)
        rc = node_lib().aslam_node_odom(self._h, _ptr(m, ctypes.c_double), float(np.float32(dt)))
        if rc < 0:
            raise AslamError(node_lib().aslam_node_error().decode())
        return rc

    def odom_msg_now(self, msg8, now):
        m = np.ascontiguousarray(msg8, np.float64)
        rc = node_lib().aslam_node_odom_now(self._h, _ptr(m, ctypes.c_double), float(now))
        if rc < 0:
            raise AslamError(node_lib().aslam_node_error().decode())
        return rc

    @property
    def N(self):
        return node_lib().aslam_node_dim(self._h)

    def state(self):
        n = self.N
        X, Z = np.empty(n), np.empty(n)
        a, b = ctypes.c_double(), ctypes.c_double()
        node_lib().aslam_node_get(self._h, _ptr(X, ctypes.c_double), _ptr(Z, ctypes.c_double), ctypes.byref(a), ctypes.byref(b))
        return X, Z, a.value, b.value

    def P(self):
        n = self.N
        P = np.empty((n, n))
        core = ctypes.c_void_p(node_lib().aslam_node_core(self._h))
        _chk(core_lib().aslam_get_state(core, 0, None, None, _ptr(P, ctypes.c_double)))
        return P

    def wait_list(self, cap=4096):
        r, b, c = np.empty(cap, np.float32), np.empty(cap, np.float32), np.empty(cap, np.uint32)
        k = node_lib().aslam_node_wait(self._h, _ptr(r, ctypes.c_float), _ptr(b, ctypes.c_float), _ptr(c, ctypes.c_uint32), cap)
        k = min(k, cap)
        return r[:k], b[:k], c[:k]

    def replay(self, traj, T=None):
        """Drive one message-level trajectory through the callbacks; returns poses[T,3], dims[T]."""
        T = traj.T if T is None else T
        poses = np.zeros((T, 3))
        dims = np.zeros(T, np.int32)
        for t in range(T):
            if traj.obs_new[t]:
                k = int(traj.n_obs[t])
                self.sensor_msg(traj.obs[t, :k, 0], traj.obs[t, :k, 1])
            if self.odom_msg(traj.odom[t], traj.dt[t]):
                poses[t] = self.state()[0][:3]
            dims[t] = self.N
        return poses, dims
