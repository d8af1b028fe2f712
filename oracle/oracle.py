"""ctypes wrapper for the CPU oracle (oracle/s3d_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It also knows how to load the compiled *reference* (oracle/_ref, built by `make -C oracle ref`
from the unmodified sources under /root/reference) through the same ctypes ABI bindings the
product uses (sift3d_amd/abi.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libs3d_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


def build(ref: bool = True) -> None:
    """Compile the oracle (and, if /root/reference exists, the reference) -- building the checker
    is not using it."""
    subprocess.run(["make", "-s", "-C", HERE, "port"], check=True)
    if ref and os.path.isdir("/root/reference/imutil"):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libsift3D.so"))


def load_ref():
    """The unmodified reference as a ``Sift3dLib`` (same API object the product exposes)."""
    from sift3d_amd import abi
    im = C.CDLL(os.path.join(REF_DIR, "libimutil.so"), mode=C.RTLD_GLOBAL)
    s = C.CDLL(os.path.join(REF_DIR, "libsift3D.so"), mode=C.RTLD_GLOBAL)
    reg_path = os.path.join(REF_DIR, "libreg.so")
    reg = C.CDLL(reg_path) if os.path.exists(reg_path) else None
    return abi.Sift3dLib(s, im, "reference", reg)


def _p(a, t):
    return a.ctypes.data_as(t)


class ReferenceFails(RuntimeError):
    """The reference's own call returns SIFT3D_FAILURE on this input (and the product must as well)."""


class Oracle:
    """The plain-C restatement."""

    def __init__(self):
        if not os.path.exists(PORT_SO):
            build(ref=False)
        L = self.L = C.CDLL(PORT_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]
        L.orc_gauss_taps.argtypes = [C.c_double, _f32p, C.c_int]
        L.orc_incremental_sigma.argtypes = [C.c_double, C.c_double]
        L.orc_incremental_sigma.restype = C.c_double
        L.orc_sep_fir.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p,
                                  _f32p, C.c_int, C.c_double]
        L.orc_detect.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f64p]
        L.orc_detect.restype = C.c_long
        L.orc_num_candidates.argtypes = [C.c_void_p]
        L.orc_num_candidates.restype = C.c_long
        L.orc_num_octaves.argtypes = [C.c_void_p]
        L.orc_get_candidates.argtypes = [C.c_void_p, _i32p, _i32p]
        L.orc_get_keypoints.argtypes = [C.c_void_p, _i32p, _f64p, _f32p]
        L.orc_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _f64p,
                                    _f64p, _f32p]
        L.orc_describe.argtypes = [C.c_void_p, C.c_long, _f64p, _i32p, _f64p, _f32p, _f32p, _f64p]
        L.orc_describe_volume.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f64p, C.c_long,
                                          _f64p, _i32p, _f64p, _f32p, _f32p, _f64p]
        L.orc_smooth_scale_raw.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _f64p]
        L.orc_dense.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f32p]
        L.orc_dense_rotate.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f64p, _f32p]
        L.orc_nn_match.argtypes = [_f32p, C.c_long, _f32p, C.c_long, C.c_float, _i32p]
        L.orc_eig_ori.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f64p, _f32p, C.c_double, _f32p, _f64p]
        L.orc_icos_bin.argtypes = [C.c_float, C.c_float, C.c_float, _f32p]
        L.orc_eig3.argtypes = [_f64p, _f64p, _f64p]
        self.ctx = C.c_void_p(L.orc_create())

    def __del__(self):
        try:
            self.L.orc_destroy(self.ctx)
        except Exception:
            pass

    def set_params(self, peak=0.1, corner=0.4, num_kp_levels=3, sigma_n=1.15, sigma0=1.6):
        return self.L.orc_set_params(self.ctx, peak, corner, num_kp_levels, sigma_n, sigma0)

    def gauss_taps(self, sigma: float) -> np.ndarray:
        buf = np.zeros(257, np.float32)
        w = self.L.orc_gauss_taps(sigma, _p(buf, _f32p), buf.size)
        assert w > 0
        return buf[:w].copy()

    def incremental_sigma(self, a, b):
        return self.L.orc_incremental_sigma(a, b)

    def sep_fir(self, vol: np.ndarray, taps: np.ndarray, units=(1., 1., 1.), unit=1.0) -> np.ndarray:
        """vol [nz,ny,nx] or [nz,ny,nx,nc] float32."""
        v = np.ascontiguousarray(vol, np.float32)
        nc = 1 if v.ndim == 3 else v.shape[3]
        nz, ny, nx = v.shape[:3]
        out = np.empty_like(v)
        tmp = np.empty_like(v)
        u = np.asarray(units, np.float64)
        t = np.ascontiguousarray(taps, np.float32)
        rc = self.L.orc_sep_fir(_p(v, _f32p), _p(out, _f32p), _p(tmp, _f32p), nx, ny, nz, nc,
                                _p(u, _f64p), _p(t, _f32p), t.size, unit)
        if rc != 0:
            raise RuntimeError("orc_sep_fir failed")
        return out

    def detect(self, vol: np.ndarray, units=(1., 1., 1.)):
        """Returns (xyzos int32 [K,5], sd float64 [K], R float32 [K,3,3])."""
        v = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = v.shape
        u = np.asarray(units, np.float64)
        k = self.L.orc_detect(self.ctx, _p(v, _f32p), nx, ny, nz, _p(u, _f64p))
        if k == -2:
            raise ReferenceFails("a NaN gradient in a candidate's orientation window: SIFT3D_detect_keypoints fails")
        if k < 0:
            raise RuntimeError("orc_detect failed")
        xyzos = np.zeros((k, 5), np.int32)
        sd = np.zeros(k, np.float64)
        R = np.zeros((k, 3, 3), np.float32)
        if k:
            self.L.orc_get_keypoints(self.ctx, _p(xyzos, _i32p), _p(sd, _f64p), _p(R, _f32p))
        return xyzos, sd, R

    def candidates(self):
        n = self.L.orc_num_candidates(self.ctx)
        xyzos = np.zeros((n, 5), np.int32)
        keep = np.zeros(n, np.int32)
        if n:
            self.L.orc_get_candidates(self.ctx, _p(xyzos, _i32p), _p(keep, _i32p))
        return xyzos, keep

    def num_octaves(self):
        return self.L.orc_num_octaves(self.ctx)

    def level(self, which: str, o: int, s: int):
        """which in {'gss','dog'} -> (data [nz,ny,nx], units, scale)."""
        w = 0 if which == "gss" else 1
        dims = (C.c_int * 3)()
        units = np.zeros(3, np.float64)
        sc = C.c_double()
        if self.L.orc_get_level(self.ctx, w, o, s, dims, _p(units, _f64p), C.byref(sc), None) != 0:
            raise IndexError((which, o, s))
        out = np.empty((dims[2], dims[1], dims[0]), np.float32)
        self.L.orc_get_level(self.ctx, w, o, s, dims, None, None, _p(out, _f32p))
        return out, units, sc.value

    def describe(self, xyz, os_, sd, R):
        xyz = np.ascontiguousarray(xyz, np.float64)
        os_ = np.ascontiguousarray(os_, np.int32)
        sd = np.ascontiguousarray(sd, np.float64)
        R = np.ascontiguousarray(R, np.float32)
        k = xyz.shape[0]
        bins = np.zeros((k, 768), np.float32)
        xyzs = np.zeros((k, 4), np.float64)
        rc = self.L.orc_describe(self.ctx, k, _p(xyz, _f64p), _p(os_, _i32p), _p(sd, _f64p),
                                 _p(R, _f32p), _p(bins, _f32p), _p(xyzs, _f64p))
        if rc != 0:
            raise RuntimeError("orc_describe failed")
        return bins, xyzs

    def describe_volume(self, vol, units, xyz, o, sd, R):
        v = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = v.shape
        u = np.asarray(units, np.float64)
        xyz = np.ascontiguousarray(xyz, np.float64)
        o = np.ascontiguousarray(o, np.int32)
        sd = np.ascontiguousarray(sd, np.float64)
        R = np.ascontiguousarray(R, np.float32)
        k = xyz.shape[0]
        bins = np.zeros((k, 768), np.float32)
        xyzs = np.zeros((k, 4), np.float64)
        self.L.orc_describe_volume(self.ctx, _p(v, _f32p), nx, ny, nz, _p(u, _f64p), k, _p(xyz, _f64p),
                                   _p(o, _i32p), _p(sd, _f64p), _p(R, _f32p), _p(bins, _f32p),
                                   _p(xyzs, _f64p))
        return bins, xyzs

    def smooth_scale_raw(self, vol, units=(1., 1., 1.)):
        v = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = v.shape
        u = np.asarray(units, np.float64)
        out = np.empty_like(v)
        if self.L.orc_smooth_scale_raw(self.ctx, _p(v, _f32p), _p(out, _f32p), nx, ny, nz, _p(u, _f64p)):
            raise RuntimeError("orc_smooth_scale_raw failed")
        return out

    def dense(self, vol, units=(1., 1., 1.), out_units=(1., 1., 1.)):
        v = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = v.shape
        u = np.asarray(units, np.float64)
        ou = np.asarray(out_units, np.float64)
        out = np.empty((nz, ny, nx, 12), np.float32)
        if self.L.orc_dense(self.ctx, _p(v, _f32p), nx, ny, nz, _p(u, _f64p), _p(ou, _f64p), _p(out, _f32p)):
            raise RuntimeError("orc_dense failed")
        return out

    def dense_rotate(self, vol, units=(1., 1., 1.)):
        v = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = v.shape
        u = np.asarray(units, np.float64)
        out = np.empty((nz, ny, nx, 12), np.float32)
        if self.L.orc_dense_rotate(self.ctx, _p(v, _f32p), nx, ny, nz, _p(u, _f64p), _p(out, _f32p)):
            raise RuntimeError("orc_dense_rotate failed")
        return out

    def describe_window_stats(self, xyz, os_, sd, R):
        """(count, checksum) of the voxels accepted by the descriptor window test, per keypoint."""
        k = len(sd)
        cnt = np.zeros(k, np.int64)
        chk = np.zeros(k, np.uint32)
        self.L.orc_set_window_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_set_window_stats(cnt.ctypes.data, chk.ctypes.data)
        try:
            self.describe(xyz, os_, sd, R)
        finally:
            self.L.orc_set_window_stats(None, None)
        return cnt, chk

    def nn_match(self, d1, d2, nn_thresh=0.8):
        a = np.ascontiguousarray(d1, np.float32)
        b = np.ascontiguousarray(d2, np.float32)
        m = np.zeros(a.shape[0], np.int32)
        if self.L.orc_nn_match(_p(a, _f32p), a.shape[0], _p(b, _f32p), b.shape[0], nn_thresh, _p(m, _i32p)):
            raise RuntimeError("orc_nn_match failed")
        return m

    def eig_ori(self, vol, units, vc, sigma):
        v = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = v.shape
        u = np.asarray(units, np.float64)
        c = np.asarray(vc, np.float32)
        R = np.zeros(9, np.float32)
        conf = C.c_double()
        rej = self.L.orc_eig_ori(_p(v, _f32p), nx, ny, nz, _p(u, _f64p), _p(c, _f32p), sigma,
                                 _p(R, _f32p), C.byref(conf))
        return rej, R.reshape(3, 3), conf.value

    def icos_bin(self, g):
        out = np.zeros(3, np.float32)
        f = self.L.orc_icos_bin(float(g[0]), float(g[1]), float(g[2]), _p(out, _f32p))
        return f, out
