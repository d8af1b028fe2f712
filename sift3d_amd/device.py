"""ctypes bindings of the flat device C-ABI (include/s3d_device.h) and of the sift3d_amd_* extension
entry points (include/sift3d_amd.h).  Plumbing only: pointers are plain integers (device addresses,
e.g. ``torch.Tensor.data_ptr()``) or numpy buffers for host data."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi

_vp = C.c_void_p
_f32p = C.POINTER(C.c_float)


def bind_extensions(L: C.CDLL) -> None:
    P = C.POINTER
    L.sift3d_amd_detect_keypoints_dev.argtypes = [P(abi.SIFT3D), _vp, C.c_int, C.c_int, C.c_int, C.c_double,
                                                  C.c_double, C.c_double, P(abi.Keypoint_store)]
    L.sift3d_amd_extract_descriptors_dev.argtypes = [P(abi.SIFT3D), P(abi.Keypoint_store), P(_vp)]
    L.sift3d_amd_extract_dense_dev.argtypes = [P(abi.SIFT3D), _vp, C.c_int, C.c_int, C.c_int, C.c_double,
                                               C.c_double, C.c_double, P(C.c_double), _vp]
    L.sift3d_amd_gauss_dev.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_double), _f32p,
                                       C.c_int, C.c_double]
    L.sift3d_amd_download_pyramid.argtypes = [P(abi.SIFT3D), C.c_int]
    L.sift3d_amd_last_num_candidates.argtypes = [P(abi.SIFT3D)]
    L.sift3d_amd_last_num_candidates.restype = C.c_long
    L.sift3d_amd_set_stream.argtypes = [P(abi.SIFT3D), _vp]
    L.sift3d_amd_nn_match_dev.argtypes = [_vp, C.c_size_t, C.c_long, _vp, C.c_size_t, C.c_long, C.c_float,
                                          P(C.c_int), _vp]
    L.sift3d_amd_last_error.restype = C.c_char_p


class DeviceLib:
    def __init__(self, L: C.CDLL):
        self.L = L
        L.s3d_rt_last_error.restype = C.c_char_p
        L.s3d_rt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.s3d_rt_malloc.argtypes = [C.POINTER(_vp), C.c_size_t]
        L.s3d_rt_free.argtypes = [_vp]
        L.s3d_rt_h2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
        L.s3d_rt_d2h.argtypes = [_vp, _vp, C.c_size_t, _vp]
        L.s3d_rt_d2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
        L.s3d_rt_memset.argtypes = [_vp, C.c_int, C.c_size_t, _vp]
        L.s3d_rt_sync.argtypes = [_vp]
        L.s3d_rt_event_create.argtypes = [C.POINTER(_vp)]
        L.s3d_rt_event_destroy.argtypes = [_vp]
        L.s3d_rt_event_record.argtypes = [_vp, _vp]
        L.s3d_rt_event_elapsed_ms.argtypes = [_vp, _vp, C.POINTER(C.c_float)]
        L.s3d_k_absmax.argtypes = [_vp, C.c_size_t, _vp, _vp]
        L.s3d_k_scale_div.argtypes = [_vp, C.c_size_t, _vp, _vp]
        L.s3d_k_decimate2.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
        L.s3d_k_subtract.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp]
        L.s3d_k_conv_axis.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int,
                                      C.c_float, _vp]
        L.s3d_k_sep_fir_path.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int,
                                         C.c_int, _vp]
        L.s3d_k_gauss_set_chunks.argtypes = [C.c_int, C.c_int]
        L.s3d_k_gauss_set_chunks.restype = None
        L.s3d_k_gauss_set_events.argtypes = [_vp, _vp, _vp]
        L.s3d_k_gauss_set_events.restype = None
        L.s3d_k_dogmax.argtypes = [_vp, _vp, C.c_size_t, _vp, _vp]
        L.s3d_mesh_table.argtypes = [_f32p]
        L.s3d_mesh_table.restype = None

    def err(self) -> str:
        return (self.L.s3d_rt_last_error() or b"").decode()

    def check(self, rc: int, what: str = "device call") -> None:
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.err()}")

    def device_count(self) -> int:
        n = C.c_int(0)
        rc = self.L.s3d_rt_device_count(C.byref(n))
        return n.value if rc == 0 else 0

    # --- memory helpers (numpy <-> HBM) -----------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = _vp()
        self.check(self.L.s3d_rt_malloc(C.byref(p), nbytes), "s3d_rt_malloc")
        return p.value

    def free(self, p: int) -> None:
        self.L.s3d_rt_free(_vp(p))

    def upload(self, a: np.ndarray, stream=None) -> int:
        a = np.ascontiguousarray(a)
        p = self.malloc(a.nbytes)
        self.check(self.L.s3d_rt_h2d(_vp(p), _vp(a.ctypes.data), a.nbytes, _vp(stream)), "h2d")
        self.check(self.L.s3d_rt_sync(_vp(stream)), "sync")
        return p

    def download(self, p: int, shape, dtype=np.float32, stream=None) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.check(self.L.s3d_rt_d2h(_vp(out.ctypes.data), _vp(p), out.nbytes, _vp(stream)), "d2h")
        self.check(self.L.s3d_rt_sync(_vp(stream)), "sync")
        return out

    def sync(self, stream=None) -> None:
        self.check(self.L.s3d_rt_sync(_vp(stream)), "sync")

    # --- one Gaussian application on device buffers ------------------------------------------------
    def sep_fir(self, d_src: int, d_dst: int, d_tmp: int, nx, ny, nz, nc, uf, taps: np.ndarray, path=0,
                stream=None) -> None:
        t = np.ascontiguousarray(taps, np.float32)
        u = np.asarray(uf, np.float32)
        self.check(self.L.s3d_k_sep_fir_path(_vp(d_src), _vp(d_dst), _vp(d_tmp), nx, ny, nz, nc,
                                             u.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), t.size, path,
                                             _vp(stream)), "s3d_k_sep_fir_path")

    def conv_axis(self, d_src, d_dst, nx, ny, nz, nc, axis, taps, uf, stream=None) -> None:
        t = np.ascontiguousarray(taps, np.float32)
        self.check(self.L.s3d_k_conv_axis(_vp(d_src), _vp(d_dst), nx, ny, nz, nc, axis, t.ctypes.data_as(_f32p),
                                          t.size, float(uf), _vp(stream)), "s3d_k_conv_axis")

    def nn_match(self, d_a: int, na: int, d_b: int, nb: int, thr: float = 0.8, stride: int = 768,
                 stream=None) -> np.ndarray:
        """SIFT3D_nn_match on device-resident descriptor rows (sift3d_amd_nn_match_dev)."""
        m = np.empty(na, np.int32)
        rc = self.L.sift3d_amd_nn_match_dev(d_a, stride, na, d_b, stride, nb, thr,
                                            m.ctypes.data_as(C.POINTER(C.c_int)), stream)
        self.check(rc, "sift3d_amd_nn_match_dev")
        return m

    def mesh_table(self) -> np.ndarray:
        out = np.zeros(20 * 16 + 32, np.float32)      # S3D_MESH_FLOATS: face table + 32-word face LUT
        self.L.s3d_mesh_table(out.ctypes.data_as(_f32p))
        return out
