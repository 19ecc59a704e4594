"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE — see oracle/rebvio_oracle.h).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEYLINE_DTYPE = np.dtype([
    ("pos", "<f4", (2,)), ("pos_img", "<f4", (2,)), ("match_pos_img", "<f4", (2,)), ("gradient", "<f4", (2,)),
    ("match_gradient", "<f4", (2,)), ("gradient_norm", "<f4"), ("match_gradient_norm", "<f4"), ("rho", "<f4"),
    ("sigma_rho", "<f4"), ("id", "<i4"), ("id_prev", "<i4"), ("id_next", "<i4"), ("match_id", "<i4"),
    ("match_id_forward", "<i4"), ("match_id_keyframe", "<i4"), ("matches", "<u4")])
assert KEYLINE_DTYPE.itemsize == 84


class Params(C.Structure):
    _fields_ = [
        ("rows", C.c_int), ("cols", C.c_int), ("fm", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("keylines_ref", C.c_int), ("keylines_max", C.c_int),
        ("pos_neg_threshold", C.c_float), ("dog_threshold", C.c_float), ("threshold", C.c_float), ("gain", C.c_float),
        ("max_threshold", C.c_float), ("min_threshold", C.c_float),
        ("search_range", C.c_float), ("reweight_distance", C.c_float), ("match_treshold", C.c_float),
        ("min_match_threshold", C.c_uint), ("iterations", C.c_uint), ("global_min_matches_threshold", C.c_uint),
        ("pixel_uncertainty", C.c_float), ("quantile_cutoff", C.c_float), ("quantile_num_bins", C.c_int),
        ("reshape_q_abs", C.c_float),
        ("pixel_uncertainty_match", C.c_float), ("match_threshold_norm", C.c_float), ("match_threshold_angle", C.c_float),
        ("regularization_threshold", C.c_float),
        ("gyro_std_dev", C.c_float), ("gyro_bias_std_dev", C.c_float)]


class PairOut(C.Structure):
    _fields_ = [
        ("Vg", C.c_float * 3), ("P_Vg", C.c_float * 9), ("F", C.c_float), ("Xv", C.c_float * 6), ("W_Xv", C.c_float * 36),
        ("Xgv", C.c_float * 6), ("V", C.c_float * 3), ("R", C.c_float * 9), ("P_V", C.c_float * 9),
        ("sigma_rho_min", C.c_float), ("ext_ok", C.c_int), ("klm_num", C.c_int), ("kf_matches", C.c_int),
        ("reg_num", C.c_int), ("lm_accept_mask", C.c_int), ("status", C.c_int)]


class VioOut(C.Structure):
    _fields_ = [("pair", PairOut), ("orientation", C.c_float * 3), ("position", C.c_float * 3), ("K", C.c_float),
                ("g_est", C.c_float * 3), ("b_est", C.c_float * 3), ("Bg", C.c_float * 3), ("initialized", C.c_int),
                ("sab_active", C.c_int)]


def host_cpu_tag() -> str:
    """A short tag of this host's CPU (model name + feature flags): -march=native code is only valid where it was built."""
    import hashlib
    import platform
    text = platform.machine()
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith(("model name", "flags")):
                    text += ln
                if ln.startswith("flags"):
                    break
    except OSError:
        pass
    return hashlib.sha1(text.encode()).hexdigest()[:12]


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile) and return the .so path. Safe to call from several processes at once
    (the ranks of `bench.py --gpus N`): the make run is serialised by a file lock, the targets have real prerequisites (the
    second caller finds the library up to date) and the Makefile renames a finished library into place."""
    import fcntl
    tag = host_cpu_tag()
    target = ["native", f"NATIVE_TAG={tag}"] if native else []
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-s", "-C", _HERE] + target, check=True)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    sub = f"_build/native-{tag}" if native else "_build"
    return os.path.join(_HERE, sub, "librebvio_oracle.so")


_lib = None


def lib(path: str | None = None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.path.join(_HERE, "_build", "librebvio_oracle.so")
    if not os.path.exists(p):
        p = build()
    L = C.CDLL(p)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    vp = C.c_void_p
    sig = {
        "orc_default_params": (None, [C.POINTER(Params), C.c_int, C.c_int]),
        "orc_create": (vp, [C.POINTER(Params)]),
        "orc_destroy": (None, [vp]),
        "orc_set_wide_sums": (None, [vp, C.c_int]),
        "orc_scale_space": (None, [vp, fp, fp, fp, fp, fp]),
        "orc_integral_image": (None, [C.c_int, C.c_int, fp, fp]),
        "orc_box_average": (None, [C.c_int, C.c_int, C.c_int, fp, fp]),
        "orc_filter_width": (C.c_int, [vp, C.c_int, C.c_int]),
        "orc_detect": (vp, [vp, fp, C.c_uint64]),
        "orc_detector_threshold": (C.c_float, [vp]),
        "orc_detector_auto_threshold": (C.c_float, [vp]),
        "orc_detector_mask": (None, [vp, ip]),
        "orc_map_size": (C.c_int, [vp]),
        "orc_map_threshold": (C.c_float, [vp]),
        "orc_map_set_threshold": (None, [vp, C.c_float]),
        "orc_map_get_keylines": (None, [vp, vp]),
        "orc_map_set_keylines": (None, [vp, vp, C.c_int]),
        "orc_map_get_mask": (None, [vp, ip]),
        "orc_map_clone": (vp, [vp]),
        "orc_map_free": (None, [vp]),
        "orc_build_distance_field": (None, [vp, vp]),
        "orc_distance_field": (None, [vp, ip, ip]),
        "orc_rotate_keylines": (None, [vp, vp, fp]),
        "orc_estimate_quantile": (C.c_float, [vp, C.c_float, C.c_int]),
        "orc_try_vel": (C.c_float, [vp, vp, fp, C.c_float, fp, fp, fp]),
        "orc_minimize_vel": (C.c_float, [vp, vp, fp, fp, ip, fp]),
        "orc_forward_match": (C.c_int, [vp, vp]),
        "orc_ext_rot_vel": (C.c_int, [vp, fp, fp, fp, fp]),
        "orc_directed_match": (C.c_int, [vp, vp, vp, fp, fp, fp, ip, C.c_float]),
        "orc_regularize": (C.c_int, [vp]),
        "orc_update_inverse_depth": (None, [vp, fp]),
        "orc_reset_state": (None, [vp]),
        "orc_track_pair": (C.c_int, [vp, vp, vp, fp, C.c_float, C.POINTER(PairOut)]),
        "orc_vio_reset": (None, [vp, fp, fp]),
        "orc_vio_add_imu": (None, [vp, vp, C.c_uint64, fp, fp]),
        "orc_vio_step": (C.c_int, [vp, vp, vp, C.POINTER(VioOut)]),
        "orc_front_end_u8": (None, [vp, C.POINTER(C.c_uint8), fp, fp, fp]),
        "orc_ls4_reset": (None, [vp]),
        "orc_estimate_ls4_acceleration": (None, [vp, fp, fp, fp, C.c_float]),
        "orc_so3_exp": (None, [fp, fp]),
        "orc_sym6_solve": (None, [fp, fp, fp]),
        "orc_search_match": (C.c_int, [vp, vp, vp, fp, fp, fp, C.c_float]),
        "orc_test_fk": (C.c_int, [vp, vp, C.c_float]),
        "orc_calculate_fj": (C.c_float, [vp, C.c_int, fp, fp, vp, C.c_float, C.c_float, ip, fp]),
        "orc_update_inverse_depth_arlu": (None, [vp, vp, fp]),
        "orc_smooth": (None, [vp, fp, C.c_float, C.c_int, fp, ip]),
        "orc_smooth_n": (None, [vp, fp, C.c_float, C.c_int, fp, ip]),
        "orc_run_stream": (C.c_double, [vp, C.POINTER(C.c_uint8), ip, C.c_int, C.c_int, ip, ip, fp]),
        "orc_run_stream_ex": (C.c_double, [vp, C.POINTER(C.c_uint8), ip, C.c_int, C.c_int, ip, ip, fp, C.POINTER(C.c_double)]),
        "orc_stage_seconds": (None, [vp, C.POINTER(C.c_double), C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    if path is None:
        _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def default_params(rows: int, cols: int, **over) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p), rows, cols)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Map:
    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        if self.h:
            self.L.orc_map_free(self.h)
            self.h = None

    def size(self):
        return self.L.orc_map_size(self.h)

    @property
    def threshold(self):
        return self.L.orc_map_threshold(self.h)

    @threshold.setter
    def threshold(self, t):
        self.L.orc_map_set_threshold(self.h, float(t))

    def keylines(self) -> np.ndarray:
        out = np.zeros(self.size(), KEYLINE_DTYPE)
        if out.size:
            self.L.orc_map_get_keylines(self.h, out.ctypes.data)
        return out

    def set_keylines(self, kl: np.ndarray):
        kl = np.ascontiguousarray(kl, KEYLINE_DTYPE)
        self.L.orc_map_set_keylines(self.h, kl.ctypes.data, len(kl))

    def mask(self, rows, cols) -> np.ndarray:
        out = np.empty((rows, cols), np.int32)
        self.L.orc_map_get_mask(self.h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def clone(self) -> "Map":
        return Map(self.L, self.L.orc_map_clone(self.h))


class Oracle:
    """Thin object wrapper over the C API."""

    def __init__(self, params: Params, path: str | None = None):
        self.L = lib(path)
        self.p = params
        self.rows, self.cols = params.rows, params.cols
        self.h = self.L.orc_create(C.byref(params))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def set_wide_sums(self, on: bool):
        """DIAGNOSTIC, not the reference: accumulate the keyline sums of tryVel / extRotVel in double."""
        self.L.orc_set_wide_sums(self.h, int(bool(on)))

    def set_sum_order(self, order: str):
        """Order of the keyline sums of tryVel / extRotVel: "reference" (fp32, index order - what the reference computes),
        or one of two DIAGNOSTICS: "wide" (double accumulation) and "device" (the fp32 terms associated as the HIP kernels
        associate them, rebvio_oracle.cpp struct Acc: with it minimizeVel and extRotVel have to reproduce the device's bits)."""
        self.L.orc_set_wide_sums(self.h, {"reference": 0, "wide": 1, "device": 2}[order])

    # --- detection -------------------------------------------------------------------------
    def scale_space(self, img):
        img, pi = _f(img)
        outs = [np.empty((self.rows, self.cols), np.float32) for _ in range(4)]
        self.L.orc_scale_space(self.h, pi, *[o.ctypes.data_as(C.POINTER(C.c_float)) for o in outs])
        return dict(scale0=outs[0], scale1=outs[1], dog=outs[2], mag=outs[3])

    def detect(self, img, ts_us=0) -> Map:
        img, pi = _f(img)
        assert img.shape == (self.rows, self.cols)
        return Map(self.L, self.L.orc_detect(self.h, pi, ts_us))

    def front_end_u8(self, frame_u8, fx, fy, cx, cy, dist) -> np.ndarray:
        """convertTo(CV_32F, 3.0) + cv::undistort(K(fx,fy,cx,cy), D = k1,k2,p1,p2,k3)."""
        frame_u8 = np.ascontiguousarray(frame_u8, np.uint8)
        k, pk = _f(np.array([fx, fy, cx, cy], np.float32))
        d, pd = _f(np.array(dist, np.float32))
        out = np.empty(frame_u8.shape, np.float32)
        self.L.orc_front_end_u8(self.h, frame_u8.ctypes.data_as(C.POINTER(C.c_uint8)), pk, pd, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def detect_u8(self, frame_u8, ts_us=0) -> Map:
        return self.detect(frame_u8.astype(np.float32) * np.float32(3.0), ts_us)

    @property
    def threshold(self):
        return self.L.orc_detector_threshold(self.h)

    @property
    def auto_threshold(self):
        return self.L.orc_detector_auto_threshold(self.h)

    # --- tracking --------------------------------------------------------------------------
    def build_distance_field(self, m: Map):
        self.L.orc_build_distance_field(self.h, m.h)

    def distance_field(self):
        ids = np.empty((self.rows, self.cols), np.int32)
        dist = np.empty((self.rows, self.cols), np.int32)
        ip = C.POINTER(C.c_int)
        self.L.orc_distance_field(self.h, ids.ctypes.data_as(ip), dist.ctypes.data_as(ip))
        return ids, dist

    def rotate(self, m: Map, R):
        R, pr = _f(np.asarray(R).reshape(9))
        self.L.orc_rotate_keylines(self.h, m.h, pr)

    def quantile(self, m: Map, pct=0.9, bins=100):
        return self.L.orc_estimate_quantile(m.h, pct, bins)

    def try_vel(self, m: Map, vel, sigma_rho_min, residuals):
        vel, pv = _f(vel)
        assert residuals.dtype == np.float32 and residuals.flags.c_contiguous
        JtJ = np.zeros(9, np.float32)
        JtF = np.zeros(3, np.float32)
        fp = C.POINTER(C.c_float)
        s = self.L.orc_try_vel(self.h, m.h, pv, sigma_rho_min, residuals.ctypes.data_as(fp), JtJ.ctypes.data_as(fp),
                               JtF.ctypes.data_as(fp))
        return s, JtJ.reshape(3, 3), JtF

    def minimize_vel(self, m: Map, vel0=(0, 0, 0)):
        vel = np.array(vel0, np.float32)
        Rvel = np.zeros(9, np.float32)
        mask = C.c_int(0)
        srm = C.c_float(0)
        fp = C.POINTER(C.c_float)
        F = self.L.orc_minimize_vel(self.h, m.h, vel.ctypes.data_as(fp), Rvel.ctypes.data_as(fp), C.byref(mask), C.byref(srm))
        return dict(F=F, vel=vel, Rvel=Rvel.reshape(3, 3), accept_mask=mask.value, sigma_rho_min=srm.value)

    def forward_match(self, old: Map, new: Map):
        return self.L.orc_forward_match(old.h, new.h)

    def ext_rot_vel(self, vel):
        vel, pv = _f(vel)
        Wx = np.zeros(36, np.float32)
        X = np.zeros(6, np.float32)
        JtF = np.zeros(6, np.float32)
        fp = C.POINTER(C.c_float)
        ok = self.L.orc_ext_rot_vel(self.h, pv, Wx.ctypes.data_as(fp), X.ctypes.data_as(fp), JtF.ctypes.data_as(fp))
        return dict(ok=ok, Wx=Wx.reshape(6, 6), X=X, JtF=JtF)

    def directed_match(self, new: Map, old: Map, vel, Rvel, Rback, max_radius=40.0):
        vel, pv = _f(vel)
        Rvel, prv = _f(np.asarray(Rvel).reshape(9))
        Rback, prb = _f(np.asarray(Rback).reshape(9))
        kf = C.c_int(0)
        n = self.L.orc_directed_match(self.h, new.h, old.h, pv, prv, prb, C.byref(kf), max_radius)
        return n, kf.value

    def regularize(self, m: Map):
        return self.L.orc_regularize(m.h)

    def search_match(self, searched: Map, query_keyline, vel, Rvel, Rback, max_radius=40.0) -> int:
        q = np.ascontiguousarray(np.asarray(query_keyline, KEYLINE_DTYPE).reshape(1))
        vel, pv = _f(vel)
        Rvel, prv = _f(np.asarray(Rvel).reshape(9))
        Rback, prb = _f(np.asarray(Rback).reshape(9))
        return self.L.orc_search_match(self.h, searched.h, q.ctypes.data, pv, prv, prb, max_radius)

    def smooth(self, img, sigma, n=3):
        img, pi = _f(img)
        out = np.empty((self.rows, self.cols), np.float32)
        w = (C.c_int * max(n, 3))()
        if n == 3:
            self.L.orc_smooth(self.h, pi, sigma, n, out.ctypes.data_as(C.POINTER(C.c_float)), w)
        else:
            self.L.orc_smooth_n(self.h, pi, sigma, n, out.ctypes.data_as(C.POINTER(C.c_float)), w)
        return out, list(w)[:n]

    def update_inverse_depth(self, vel):
        vel, pv = _f(vel)
        self.L.orc_update_inverse_depth(self.h, pv)

    def reset_state(self):
        self.L.orc_reset_state(self.h)

    def track_pair(self, old: Map, new: Map, R_prior=None, frame_dt=0.05) -> PairOut:
        out = PairOut()
        pr = None
        if R_prior is not None:
            R_prior, pr = _f(np.asarray(R_prior).reshape(9))
        self.L.orc_track_pair(self.h, old.h, new.h, pr, frame_dt, C.byref(out))
        return out

    # --- full VIO (config 5) -------------------------------------------------------------------
    def vio_reset(self, R_c2i=None, t_c2i=None):
        R, pr = _f(np.eye(3) if R_c2i is None else np.asarray(R_c2i).reshape(9))
        t, pt = _f(np.zeros(3) if t_c2i is None else t_c2i)
        self.L.orc_vio_reset(self.h, pr, pt)

    def vio_add_imu(self, m: Map, ts_us, gyro, acc):
        g, pg = _f(gyro)
        a, pa = _f(acc)
        self.L.orc_vio_add_imu(self.h, m.h, int(ts_us), pg, pa)

    def vio_step(self, old: Map, new: Map) -> VioOut:
        out = VioOut()
        self.L.orc_vio_step(self.h, old.h, new.h, C.byref(out))
        return out

    def run_stream(self, frames_u8: np.ndarray, idx: np.ndarray, threads=1):
        frames_u8 = np.ascontiguousarray(frames_u8, np.uint8)
        idx = np.ascontiguousarray(idx, np.int32)
        n = len(idx)
        kc = np.zeros(n, np.int32)
        mc = np.zeros(n, np.int32)
        pose = np.zeros((n, 6), np.float32)
        done = np.zeros(n, np.float64)
        ip = C.POINTER(C.c_int)
        self.stage_seconds(reset=True)
        secs = self.L.orc_run_stream_ex(self.h, frames_u8.ctypes.data_as(C.POINTER(C.c_uint8)), idx.ctypes.data_as(ip), n, threads,
                                        kc.ctypes.data_as(ip), mc.ctypes.data_as(ip), pose.ctypes.data_as(C.POINTER(C.c_float)),
                                        done.ctypes.data_as(C.POINTER(C.c_double)))
        return dict(seconds=secs, frames=n, keyline_counts=kc, match_counts=mc, pose=pose, frame_done_s=done,
                    stage_seconds=self.stage_seconds())

    STAGES = ("detect", "buildDistanceField", "minimizeVel", "extRotVel", "directedMatch", "other_track")

    def stage_seconds(self, reset=False) -> dict:
        """Wall seconds accumulated at the reference's REBVIO_TIMER tick sites (rebvio_oracle.h: orc_stage_seconds)."""
        out = (C.c_double * 6)()
        self.L.orc_stage_seconds(self.h, out, int(bool(reset)))
        return dict(zip(self.STAGES, [float(v) for v in out]))
