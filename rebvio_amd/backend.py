"""ctypes binding of the gfx950 backend's C-ABI (include/rebvio_hip.h).

Plumbing only: this module loads `rebvio_amd/_build/librebvio_hip.so` and forwards calls. There is no CPU
fallback anywhere: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# REBVIO_HIP_LIB: another build of the same library (A/B measurements of compile-time variants); default = the in-tree build
LIB_PATH = os.environ.get("REBVIO_HIP_LIB") or os.path.join(_HERE, "_build", "librebvio_hip.so")

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
        ("gyro_std_dev", C.c_float), ("gyro_bias_std_dev", C.c_float),
        ("device_id", C.c_int), ("map_pool", C.c_int)]


class PairOut(C.Structure):
    _fields_ = [
        ("Vg", C.c_float * 3), ("P_Vg", C.c_float * 9), ("F", C.c_float), ("Xv", C.c_float * 6), ("W_Xv", C.c_float * 36),
        ("Xgv", C.c_float * 6), ("V", C.c_float * 3), ("R", C.c_float * 9), ("P_V", C.c_float * 9),
        ("sigma_rho_min", C.c_float), ("ext_ok", C.c_int), ("klm_num", C.c_int), ("kf_matches", C.c_int),
        ("reg_num", C.c_int), ("lm_accept_mask", C.c_int), ("status", C.c_int)]


class PairMid(C.Structure):
    _fields_ = [("Vg", C.c_float * 3), ("P_Vg", C.c_float * 9), ("F", C.c_float), ("sigma_rho_min", C.c_float),
                ("lm_accept_mask", C.c_int), ("ext_ok", C.c_int), ("Xv", C.c_float * 6), ("W_Xv", C.c_float * 36),
                ("Xgv", C.c_float * 6), ("W_Xgv", C.c_float * 36), ("R", C.c_float * 9)]


# every symbol include/rebvio_hip.h declares: (restype, argtypes)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
SIGNATURES = {
    "rebvio_hip_abi_version": (C.c_int, []),
    "rebvio_hip_last_error": (C.c_char_p, []),
    "rebvio_hip_default_params": (None, [C.POINTER(Params), C.c_int, C.c_int]),
    "rebvio_hip_create": (C.c_int, [C.POINTER(Params), C.POINTER(_vp)]),
    "rebvio_hip_destroy": (None, [_vp]),
    "rebvio_hip_scale_space": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp]),
    "rebvio_hip_detect": (C.c_int, [_vp, _fp, C.c_size_t, C.c_uint64, C.POINTER(_vp)]),
    "rebvio_hip_detect_u8_device": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(_vp)]),
    "rebvio_hip_detect_u8": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.POINTER(_vp)]),
    "rebvio_hip_set_undistort": (C.c_int, [_vp, _fp, _fp]),
    "rebvio_hip_front_end_u8": (C.c_int, [_vp, _vp, _fp]),
    "rebvio_hip_detector_state": (C.c_int, [_vp, _fp, _fp, _ip]),
    "rebvio_hip_map_size": (C.c_int, [_vp]),
    "rebvio_hip_map_threshold": (C.c_float, [_vp]),
    "rebvio_hip_map_ts": (C.c_uint64, [_vp]),
    "rebvio_hip_map_download": (C.c_int, [_vp, _vp, _ip]),
    "rebvio_hip_render_edge_image": (C.c_int, [_vp, _vp, _vp]),
    "rebvio_hip_map_upload": (C.c_int, [_vp, _vp, C.c_int]),
    "rebvio_hip_map_release": (None, [_vp]),
    "rebvio_hip_build_distance_field": (C.c_int, [_vp, _vp]),
    "rebvio_hip_distance_field": (C.c_int, [_vp, _ip, _ip]),
    "rebvio_hip_map_distance_field": (C.c_int, [_vp, _ip, _ip]),
    "rebvio_hip_search_match": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _fp, C.c_float, _ip]),
    "rebvio_hip_smooth": (C.c_int, [_vp, _fp, _ip, _fp]),
    "rebvio_hip_smooth_n": (C.c_int, [_vp, _fp, _ip, C.c_int, _fp]),
    "rebvio_hip_rotate": (C.c_int, [_vp, _vp, _fp]),
    "rebvio_hip_quantile": (C.c_int, [_vp, _vp, C.c_float, C.c_int, _fp]),
    "rebvio_hip_try_vel": (C.c_int, [_vp, _vp, _fp, C.c_float, _fp, _fp]),
    "rebvio_hip_minimize_vel": (C.c_int, [_vp, _vp, _fp, _fp, _fp, _ip, _fp]),
    "rebvio_hip_forward_match": (C.c_int, [_vp, _vp, _vp]),
    "rebvio_hip_ext_rot_vel": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _ip]),
    "rebvio_hip_directed_match": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _fp, C.c_float, _ip, _ip]),
    "rebvio_hip_regularize": (C.c_int, [_vp, _vp, _ip]),
    "rebvio_hip_update_inverse_depth": (C.c_int, [_vp, _fp]),
    "rebvio_hip_reset_state": (None, [_vp]),
    "rebvio_hip_get_gyro_state": (C.c_int, [_vp, _fp, _fp]),
    "rebvio_hip_set_gyro_state": (C.c_int, [_vp, _fp, _fp]),
    "rebvio_hip_track_pair": (C.c_int, [_vp, _vp, _vp, _fp, C.c_float, C.POINTER(PairOut)]),
    "rebvio_hip_track_pair_begin": (C.c_int, [_vp, _vp, _vp, _fp, C.c_float, C.POINTER(PairMid)]),
    "rebvio_hip_track_pair_finish": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _fp, _fp, _ip, _ip, _ip, _ip]),
    "rebvio_hip_track_pair_finish_async": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _fp, _fp, _fp]),
    "rebvio_hip_track_pair_result": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "rebvio_hip_track_pair_hint_next": (C.c_int, [_vp, _vp]),
    "rebvio_hip_push_frame_u8_device": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(PairOut), _ip]),
    "rebvio_hip_push_frame_u8": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.POINTER(PairOut), _ip]),
    "rebvio_hip_next_record": (C.c_int, [_vp, C.POINTER(PairOut), _ip]),
    "rebvio_hip_pairs_started": (C.c_uint64, [_vp]),
    "rebvio_hip_flush": (C.c_int, [_vp]),
    "rebvio_hip_batch_create": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(_vp)]),
    "rebvio_hip_batch_destroy": (None, [_vp]),
    "rebvio_hip_batch_lanes": (C.c_int, [_vp]),
    "rebvio_hip_batch_lane": (_vp, [_vp, C.c_int]),
    "rebvio_hip_batch_push_u8_device": (C.c_int, [_vp, C.POINTER(_vp), C.c_uint64, C.POINTER(PairOut), _ip]),
    "rebvio_hip_batch_next_records": (C.c_int, [_vp, C.POINTER(PairOut), _ip]),
    "rebvio_hip_batch_flush": (C.c_int, [_vp]),
    "rebvio_hip_test_glue": (C.c_int, [_vp, _fp, _fp, C.c_float, C.c_float, C.c_int, _fp, C.c_int, C.c_float, _fp, _fp, _fp,
                                      C.POINTER(PairOut), _fp, _fp, C.POINTER(PairOut), _fp, _fp]),
    "rebvio_hip_test_forge_record_stamp": (C.c_int, [_vp]),
    "rebvio_hip_batch_test_forge_record_stamp": (C.c_int, [_vp]),
    "rebvio_hip_profile_enable": (C.c_int, [_vp, C.c_int]),
    "rebvio_hip_profile_select": (C.c_int, [_vp, C.c_char_p]),
    "rebvio_hip_profile_reset": (C.c_int, [_vp]),
    "rebvio_hip_profile_read": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), _ip, C.c_int]),
    "rebvio_hip_device_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "rebvio_hip_device_free": (C.c_int, [_vp, _vp]),
    "rebvio_hip_device_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
}


def build(verbose: bool = False) -> str:
    """Compile the backend for gfx950 with hipcc (rebvio_amd/csrc/Makefile)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")] + ([] if verbose else ["-s"]), check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP backend has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class HipError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise HipError(f"rebvio_hip error {rc}: {lib().rebvio_hip_last_error().decode()}")


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(_fp)


def default_params(rows, cols, **over) -> Params:
    p = Params()
    lib().rebvio_hip_default_params(C.byref(p), rows, cols)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Map:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def release(self):
        if self.h:
            lib().rebvio_hip_map_release(self.h)  # safe after the context is closed too: the library drops the husk
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def size(self):
        n = lib().rebvio_hip_map_size(self.h)
        if n < 0:
            raise HipError(lib().rebvio_hip_last_error().decode())
        return n

    @property
    def threshold(self):
        return lib().rebvio_hip_map_threshold(self.h)

    def render_edge_image(self, gray_u8=None) -> np.ndarray:
        rows, cols = self.ctx.rows, self.ctx.cols
        out = np.empty((rows, cols, 3), np.uint8)
        g = None
        if gray_u8 is not None:
            gray_u8 = np.ascontiguousarray(gray_u8, np.uint8)
            assert gray_u8.shape == (rows, cols)
            g = gray_u8.ctypes.data_as(_vp)
        _chk(lib().rebvio_hip_render_edge_image(self.h, g, out.ctypes.data_as(_vp)))
        return out

    def keylines(self) -> np.ndarray:
        out = np.zeros(self.size(), KEYLINE_DTYPE)
        _chk(lib().rebvio_hip_map_download(self.h, out.ctypes.data if out.size else None, None))
        return out

    def mask(self) -> np.ndarray:
        out = np.empty((self.ctx.rows, self.ctx.cols), np.int32)
        _chk(lib().rebvio_hip_map_download(self.h, None, out.ctypes.data_as(_ip)))
        return out

    def distance_field(self):
        """The field built from THIS map (DistanceField::operator[] for all cells)."""
        ids = np.empty((self.ctx.rows, self.ctx.cols), np.int32)
        dist = np.empty((self.ctx.rows, self.ctx.cols), np.int32)
        _chk(lib().rebvio_hip_map_distance_field(self.h, ids.ctypes.data_as(_ip), dist.ctypes.data_as(_ip)))
        return ids, dist

    def search_match(self, query_keyline, vel, Rvel, Rback, max_radius=40.0) -> int:
        """EdgeMap::searchMatch: this map is searched for a match of `query_keyline` (one KEYLINE_DTYPE record)."""
        q = np.ascontiguousarray(np.asarray(query_keyline, KEYLINE_DTYPE).reshape(1))
        vel, pv = _f(vel)
        Rvel, prv = _f(np.asarray(Rvel).reshape(9))
        Rback, prb = _f(np.asarray(Rback).reshape(9))
        out = C.c_int(-2)
        _chk(lib().rebvio_hip_search_match(self.ctx.h, self.h, q.ctypes.data, pv, prv, prb, max_radius, C.byref(out)))
        return out.value

    def upload(self, kl: np.ndarray):
        kl = np.ascontiguousarray(kl, KEYLINE_DTYPE)
        _chk(lib().rebvio_hip_map_upload(self.h, kl.ctypes.data if kl.size else None, len(kl)))


class Context:
    def __init__(self, params: Params, _handle=None):
        self.p = params
        self.rows, self.cols = params.rows, params.cols
        self._owned = _handle is None
        if _handle is None:
            h = _vp()
            _chk(lib().rebvio_hip_create(C.byref(params), C.byref(h)))
        else:
            h = _vp(_handle)
        self.h = h
        self._dev_bufs = []

    def close(self):
        if getattr(self, "h", None):
            for b in self._dev_bufs:
                lib().rebvio_hip_device_free(self.h, b)
            self._dev_bufs = []
            if self._owned:
                lib().rebvio_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # detection
    def scale_space(self, img):
        img, pi = _f(img)
        outs = [np.empty((self.rows, self.cols), np.float32) for _ in range(4)]
        _chk(lib().rebvio_hip_scale_space(self.h, pi, *[o.ctypes.data_as(_fp) for o in outs]))
        return dict(scale0=outs[0], scale1=outs[1], dog=outs[2], mag=outs[3])

    def smooth(self, img, widths):
        """FastGaussian::smooth with the given box widths (three for the reference's own filters, any count for its general n)."""
        img, pi = _f(img)
        n = len(widths)
        w = (C.c_int * n)(*[int(v) for v in widths])
        out = np.empty((self.rows, self.cols), np.float32)
        if n == 3:
            _chk(lib().rebvio_hip_smooth(self.h, pi, w, out.ctypes.data_as(_fp)))
        else:
            _chk(lib().rebvio_hip_smooth_n(self.h, pi, w, n, out.ctypes.data_as(_fp)))
        return out

    def detect(self, img, ts_us=0) -> Map:
        img, pi = _f(img)
        assert img.shape == (self.rows, self.cols)
        h = _vp()
        _chk(lib().rebvio_hip_detect(self.h, pi, 0, ts_us, C.byref(h)))
        return Map(self, h)

    def detect_u8(self, frame_u8, ts_us=0) -> Map:
        return self.detect(frame_u8.astype(np.float32) * np.float32(3.0), ts_us)

    def detect_u8_host(self, frame_u8, ts_us=0) -> Map:
        """u8 host frame through the device front end (x3, undistort when a lens model is set)."""
        frame_u8 = np.ascontiguousarray(frame_u8, np.uint8)
        assert frame_u8.shape == (self.rows, self.cols)
        h = _vp()
        _chk(lib().rebvio_hip_detect_u8(self.h, frame_u8.ctypes.data_as(_vp), 0, ts_us, C.byref(h)))
        return Map(self, h)

    def set_undistort(self, fx, fy, cx, cy, dist):
        k, pk = _f(np.array([fx, fy, cx, cy], np.float32))
        d, pd = _f(np.array(dist, np.float32))
        assert d.size == 5
        _chk(lib().rebvio_hip_set_undistort(self.h, pk, pd))

    def front_end_u8(self, frame_u8) -> np.ndarray:
        frame_u8 = np.ascontiguousarray(frame_u8, np.uint8)
        assert frame_u8.shape == (self.rows, self.cols)
        out = np.empty((self.rows, self.cols), np.float32)
        _chk(lib().rebvio_hip_front_end_u8(self.h, frame_u8.ctypes.data_as(_vp), out.ctypes.data_as(_fp)))
        return out

    def upload_frames(self, frames_u8: np.ndarray) -> int:
        """Stage u8 frames in HBM; returns the device address of frame 0."""
        frames_u8 = np.ascontiguousarray(frames_u8, np.uint8)
        d = _vp()
        _chk(lib().rebvio_hip_device_alloc(self.h, frames_u8.nbytes, C.byref(d)))
        _chk(lib().rebvio_hip_device_upload(self.h, d, frames_u8.ctypes.data, frames_u8.nbytes))
        self._dev_bufs.append(d)
        return d.value

    def detect_u8_device(self, dev_addr: int, ts_us=0) -> Map:
        h = _vp()
        _chk(lib().rebvio_hip_detect_u8_device(self.h, _vp(dev_addr), ts_us, C.byref(h)))
        return Map(self, h)

    def detector_state(self):
        t, a, n = C.c_float(), C.c_float(), C.c_int()
        _chk(lib().rebvio_hip_detector_state(self.h, C.byref(t), C.byref(a), C.byref(n)))
        return t.value, a.value, n.value

    # tracking
    def build_distance_field(self, m: Map):
        _chk(lib().rebvio_hip_build_distance_field(self.h, m.h))

    def distance_field(self):
        ids = np.empty((self.rows, self.cols), np.int32)
        dist = np.empty((self.rows, self.cols), np.int32)
        _chk(lib().rebvio_hip_distance_field(self.h, ids.ctypes.data_as(_ip), dist.ctypes.data_as(_ip)))
        return ids, dist

    def rotate(self, m: Map, R):
        R, pr = _f(np.asarray(R).reshape(9))
        _chk(lib().rebvio_hip_rotate(self.h, m.h, pr))

    def quantile(self, m: Map, pct=0.9, bins=100):
        out = C.c_float()
        _chk(lib().rebvio_hip_quantile(self.h, m.h, pct, bins, C.byref(out)))
        return out.value

    def try_vel(self, m: Map, vel, sigma_rho_min, residuals):
        vel, pv = _f(vel)
        assert residuals.dtype == np.float32 and residuals.flags.c_contiguous
        out = np.zeros(10, np.float32)
        _chk(lib().rebvio_hip_try_vel(self.h, m.h, pv, sigma_rho_min, residuals.ctypes.data_as(_fp), out.ctypes.data_as(_fp)))
        J = np.array([[out[1], out[4], out[5]], [out[4], out[2], out[6]], [out[5], out[6], out[3]]], np.float32)
        return float(out[0]), J, out[7:10].copy()

    def minimize_vel(self, m: Map, vel0=(0, 0, 0)):
        vel = np.array(vel0, np.float32)
        Rvel = np.zeros(9, np.float32)
        F, srm, mask = C.c_float(), C.c_float(), C.c_int()
        _chk(lib().rebvio_hip_minimize_vel(self.h, m.h, vel.ctypes.data_as(_fp), Rvel.ctypes.data_as(_fp), C.byref(F),
                                           C.byref(mask), C.byref(srm)))
        return dict(F=F.value, vel=vel, Rvel=Rvel.reshape(3, 3), accept_mask=mask.value, sigma_rho_min=srm.value)

    def forward_match(self, old: Map, new: Map):
        _chk(lib().rebvio_hip_forward_match(self.h, old.h, new.h))

    def ext_rot_vel(self, vel):
        vel, pv = _f(vel)
        Wx = np.zeros(36, np.float32)
        JtF = np.zeros(6, np.float32)
        X = np.zeros(6, np.float32)
        ok = C.c_int()
        _chk(lib().rebvio_hip_ext_rot_vel(self.h, pv, Wx.ctypes.data_as(_fp), JtF.ctypes.data_as(_fp), X.ctypes.data_as(_fp),
                                          C.byref(ok)))
        return dict(ok=ok.value, Wx=Wx.reshape(6, 6), X=X, JtF=JtF)

    def directed_match(self, new: Map, old: Map, vel, Rvel, Rback, max_radius=40.0):
        vel, pv = _f(vel)
        Rvel, prv = _f(np.asarray(Rvel).reshape(9))
        Rback, prb = _f(np.asarray(Rback).reshape(9))
        n, kf = C.c_int(), C.c_int()
        _chk(lib().rebvio_hip_directed_match(self.h, new.h, old.h, pv, prv, prb, max_radius, C.byref(n), C.byref(kf)))
        return n.value, kf.value

    def regularize(self, m: Map):
        n = C.c_int()
        _chk(lib().rebvio_hip_regularize(self.h, m.h, C.byref(n)))
        return n.value

    def update_inverse_depth(self, vel):
        vel, pv = _f(vel)
        _chk(lib().rebvio_hip_update_inverse_depth(self.h, pv))

    def reset_state(self):
        lib().rebvio_hip_reset_state(self.h)

    def gyro_state(self):
        """(Bg[3], W_Bg[3, 3]) of the pair glue's gyro-bias filter (types/imu.hpp:180-182)."""
        bg = np.zeros(3, np.float32)
        w = np.zeros(9, np.float32)
        _chk(lib().rebvio_hip_get_gyro_state(self.h, bg.ctypes.data_as(_fp), w.ctypes.data_as(_fp)))
        return bg, w.reshape(3, 3)

    def set_gyro_state(self, bg, w_bg):
        bg, pb = _f(np.asarray(bg).reshape(3))
        w, pw = _f(np.asarray(w_bg).reshape(9))
        _chk(lib().rebvio_hip_set_gyro_state(self.h, pb, pw))

    def track_pair(self, old: Map, new: Map, R_prior=None, frame_dt=0.05) -> PairOut:
        out = PairOut()
        pr = None
        if R_prior is not None:
            R_prior, pr = _f(np.asarray(R_prior).reshape(9))
        _chk(lib().rebvio_hip_track_pair(self.h, old.h, new.h, pr, frame_dt, C.byref(out)))
        return out

    # the pair step in two halves (what rebvio::Rebvio drives, with the host's inertial fusion in between)
    def track_pair_begin(self, old: Map, new: Map, R_prior=None, frame_dt=0.05) -> PairMid:
        mid = PairMid()
        pr = None
        if R_prior is not None:
            R_prior, pr = _f(np.asarray(R_prior).reshape(9))
        _chk(lib().rebvio_hip_track_pair_begin(self.h, old.h, new.h, pr, frame_dt, C.byref(mid)))
        return mid

    def track_pair_finish_async(self, old: Map, new: Map, V, P_V, Rgva, R_second, R_prior_next=None):
        V, pv = _f(V)
        P_V, ppv = _f(np.asarray(P_V).reshape(9))
        Rgva, prg = _f(np.asarray(Rgva).reshape(9))
        R_second, pr2 = _f(np.asarray(R_second).reshape(9))
        prn = None
        if R_prior_next is not None:
            R_prior_next, prn = _f(np.asarray(R_prior_next).reshape(9))
        _chk(lib().rebvio_hip_track_pair_finish_async(self.h, old.h, new.h, pv, ppv, prg, pr2, prn))

    def track_pair_result(self):
        """(klm_num, kf_matches, reg_num, status) of the pair whose _finish_async came last but one / last."""
        v = [C.c_int() for _ in range(4)]
        _chk(lib().rebvio_hip_track_pair_result(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def track_pair_hint_next(self, next_new: Map):
        _chk(lib().rebvio_hip_track_pair_hint_next(self.h, next_new.h))

    def push_frame_u8_device(self, dev_addr: int, ts_us: int):
        out = PairOut()
        n = C.c_int()
        _chk(lib().rebvio_hip_push_frame_u8_device(self.h, _vp(dev_addr), ts_us, C.byref(out), C.byref(n)))
        return out, n.value

    def push_frame_u8(self, frame_u8: np.ndarray, ts_us: int):
        """Streaming push of a MONO8 frame in host memory (rows x cols, C-contiguous rows; a row pitch is taken from the array)."""
        assert frame_u8.dtype == np.uint8 and frame_u8.shape == (self.rows, self.cols) and frame_u8.strides[1] == 1
        out = PairOut()
        n = C.c_int()
        _chk(lib().rebvio_hip_push_frame_u8(self.h, _vp(frame_u8.ctypes.data), frame_u8.strides[0], ts_us, C.byref(out), C.byref(n)))
        return out, n.value

    def test_glue(self, vel, JtJ6, F, sigma_rho_min, accept_mask, xrv, n_new, frame_dt, Bg, W_Bg, R_prior):
        """The pair glue on the device and on the host from the same inputs: ((out, state[22], second[44 words]) per side)."""
        vel, pv = _f(vel)
        JtJ6, pj = _f(JtJ6)
        xrv, px = _f(np.asarray(xrv, np.float32).reshape(-1))
        Bg, pb = _f(Bg)
        W_Bg, pw = _f(np.asarray(W_Bg).reshape(9))
        R_prior, pr = _f(np.asarray(R_prior).reshape(9))
        res = []
        outs = [PairOut(), PairOut()]
        sts = [np.zeros(22, np.float32) for _ in range(2)]
        sec = [np.zeros(44, np.float32) for _ in range(2)]
        _chk(lib().rebvio_hip_test_glue(self.h, pv, pj, F, sigma_rho_min, accept_mask, px, n_new, frame_dt, pb, pw, pr,
                                        C.byref(outs[0]), sts[0].ctypes.data_as(_fp), sec[0].ctypes.data_as(_fp),
                                        C.byref(outs[1]), sts[1].ctypes.data_as(_fp), sec[1].ctypes.data_as(_fp)))
        for i in range(2):
            res.append((outs[i], sts[i], sec[i]))
        return res

    def pairs_started(self) -> int:
        """Frame pairs the streaming driver has queued on the device so far."""
        return int(lib().rebvio_hip_pairs_started(self.h))

    def next_record(self):
        """The oldest complete pair record not handed out yet, or None."""
        out = PairOut()
        n = C.c_int()
        k = lib().rebvio_hip_next_record(self.h, C.byref(out), C.byref(n))
        if k < 0:
            _chk(k)
        return (out, n.value) if k == 1 else None

    def flush(self):
        """Finishes every pair in flight; returns the records that were still waiting to be handed out, oldest first."""
        _chk(lib().rebvio_hip_flush(self.h))
        rest = []
        while True:
            r = self.next_record()
            if r is None:
                return rest
            rest.append(r)

    # profiling
    def profile(self, on: bool, only: str | None = None, stride: int = 1):
        lib().rebvio_hip_profile_select(self.h, (only or "").encode())
        lib().rebvio_hip_profile_enable(self.h, (max(1, stride) if on else 0))

    def profile_reset(self):
        lib().rebvio_hip_profile_reset(self.h)

    def profile_read(self):
        cap = 64
        names = C.create_string_buffer(8192)
        avg = (C.c_double * cap)()
        calls = (C.c_int * cap)()
        k = lib().rebvio_hip_profile_read(self.h, names, 8192, avg, calls, cap)
        ns = names.value.decode().split("\n")
        return {ns[i]: (avg[i], calls[i]) for i in range(k)}


class Batch:
    """`lanes` camera streams of one GPU advanced in lock-step (rebvio_hip_batch_*)."""

    def __init__(self, params: Params, lanes: int):
        self.p, self.B = params, lanes
        h = _vp()
        _chk(lib().rebvio_hip_batch_create(C.byref(params), lanes, C.byref(h)))
        self.h = h
        self.lanes = [Context(params, _handle=lib().rebvio_hip_batch_lane(h, l)) for l in range(lanes)]
        self._frames = (_vp * lanes)()
        self._out = (PairOut * lanes)()
        self._n = (C.c_int * lanes)()

    def push_u8_device(self, dev_addrs, ts_us: int):
        for l, a in enumerate(dev_addrs):
            self._frames[l] = int(a)
        _chk(lib().rebvio_hip_batch_push_u8_device(self.h, self._frames, ts_us, self._out, self._n))
        return self._out, self._n

    def flush(self):
        """Finishes every step in flight; returns the steps' records that were still waiting, oldest first, as
        (list of PairOut per lane, list of keyline counts per lane)."""
        _chk(lib().rebvio_hip_batch_flush(self.h))
        rest = []
        while True:
            out = (PairOut * self.B)()
            n = (C.c_int * self.B)()
            k = lib().rebvio_hip_batch_next_records(self.h, out, n)
            if k < 0:
                _chk(k)
            if k != 1:
                return rest
            rest.append((list(out), list(n)))

    def close(self):
        if getattr(self, "h", None):
            for c in self.lanes:
                c.close()  # frees the frames staged through the lane; the lane itself belongs to the batch
            lib().rebvio_hip_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
