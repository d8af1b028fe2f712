"""ctypes mirror of the SIFT3D C ABI (x86-64 SysV) that this repo's drop-in library exports.

The struct layouts below are the ones declared in ``include/sift3d_abi.h`` (which is layout
compatible with the reference's ``imutil/imtypes.h:136-334`` -- sizes/offsets are asserted both
in the header and in ``tests/test_abi.py``).  Because the layouts are identical, the very same
bindings can drive

* ``sift3d_amd/lib/libsift3d_amd.so``   -- the MI355X/HIP implementation (the product), and
* ``oracle/_ref/libsift3D.so``          -- the unmodified reference compiled as an oracle
                                           (tests / cpu_baseline only),

which is what lets the parity tests read like the reference's own usage
(``examples/featuresC.c:26-102``): init_SIFT3D -> SIFT3D_detect_keypoints ->
SIFT3D_extract_descriptors.

Nothing in here computes anything; it is plumbing.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

IM_NDIMS = 3
ICOS_NFACES = 20
ICOS_NVERT = 12
NHIST_PER_DIM = 4
HIST_NUMEL = ICOS_NVERT
DESC_NUM_TOTAL_HIST = NHIST_PER_DIM ** 3
DESC_NUMEL = DESC_NUM_TOTAL_HIST * HIST_NUMEL

SIFT3D_SUCCESS = 0
SIFT3D_FAILURE = -1


class Mat_rm(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_size_t), ("num_cols", C.c_int),
                ("num_rows", C.c_int), ("static_mem", C.c_int), ("type", C.c_int)]


class Image(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("cl_image", C.c_int), ("s", C.c_double),
                ("size", C.c_size_t), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("ux", C.c_double), ("uy", C.c_double), ("uz", C.c_double),
                ("xs", C.c_size_t), ("ys", C.c_size_t), ("zs", C.c_size_t),
                ("nc", C.c_int), ("cl_valid", C.c_int)]


class Sep_FIR_filter(C.Structure):
    _fields_ = [("cl_apply_unrolled", C.c_int), ("kernel", C.POINTER(C.c_float)),
                ("dim", C.c_int), ("width", C.c_int), ("symmetric", C.c_int)]


class Gauss_filter(C.Structure):
    _fields_ = [("sigma", C.c_double), ("f", Sep_FIR_filter)]


class GSS_filters(C.Structure):
    _fields_ = [("first_gauss", Gauss_filter), ("gauss_octave", C.POINTER(Gauss_filter)),
                ("num_filters", C.c_int), ("first_level", C.c_int)]


class SIFT_cl_kernels(C.Structure):
    _fields_ = [("downsample_2", C.c_int)]


class Pyramid(C.Structure):
    _fields_ = [("levels", C.POINTER(Image)), ("sigma_n", C.c_double), ("sigma0", C.c_double),
                ("num_kp_levels", C.c_int), ("first_octave", C.c_int), ("num_octaves", C.c_int),
                ("first_level", C.c_int), ("num_levels", C.c_int)]


class Cvec(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Slab(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("num", C.c_size_t), ("buf_size", C.c_size_t)]


class Keypoint(C.Structure):
    _fields_ = [("r_data", C.c_float * 9), ("R", Mat_rm), ("xd", C.c_double), ("yd", C.c_double),
                ("zd", C.c_double), ("sd", C.c_double), ("o", C.c_int), ("s", C.c_int)]


class Keypoint_store(C.Structure):
    _fields_ = [("buf", C.POINTER(Keypoint)), ("slab", Slab), ("nx", C.c_int), ("ny", C.c_int),
                ("nz", C.c_int)]


class Hist(C.Structure):
    _fields_ = [("bins", C.c_float * HIST_NUMEL)]


class SIFT3D_Descriptor(C.Structure):
    _fields_ = [("hists", Hist * DESC_NUM_TOTAL_HIST), ("xd", C.c_double), ("yd", C.c_double),
                ("zd", C.c_double), ("sd", C.c_double)]


class SIFT3D_Descriptor_store(C.Structure):
    _fields_ = [("buf", C.POINTER(SIFT3D_Descriptor)), ("num", C.c_size_t), ("nx", C.c_int),
                ("ny", C.c_int), ("nz", C.c_int)]


class Tri(C.Structure):
    _fields_ = [("v", Cvec * 3), ("idx", C.c_int * 3)]


class Mesh(C.Structure):
    _fields_ = [("tri", C.POINTER(Tri)), ("num", C.c_int)]


class SIFT3D(C.Structure):
    _fields_ = [("mesh", Mesh), ("gss", GSS_filters), ("kernels", SIFT_cl_kernels),
                ("gpyr", Pyramid), ("dog", Pyramid), ("im", Image), ("peak_thresh", C.c_double),
                ("corner_thresh", C.c_double), ("dense_rotate", C.c_int)]


class Ransac(C.Structure):
    _fields_ = [("err_thresh", C.c_double), ("num_iter", C.c_int)]


class Tform(C.Structure):
    _fields_ = [("type", C.c_int), ("vtable", C.c_void_p)]


class Affine(C.Structure):
    _fields_ = [("tform", Tform), ("A", Mat_rm)]


class Reg_SIFT3D(C.Structure):
    _fields_ = [("src_units", C.c_double * 3), ("ref_units", C.c_double * 3), ("sift3d", SIFT3D), ("ran", Ransac),
                ("desc_src", SIFT3D_Descriptor_store), ("desc_ref", SIFT3D_Descriptor_store), ("match_src", Mat_rm),
                ("match_ref", Mat_rm), ("nn_thresh", C.c_double), ("verbose", C.c_int)]


# (struct, sizeof, {field: offset}) measured on the compiled reference.
ABI_LAYOUT = [
    (Image, 104, {"data": 0, "cl_image": 8, "s": 16, "size": 24, "nx": 32, "ux": 48, "xs": 72,
                  "nc": 96, "cl_valid": 100}),
    (Mat_rm, 32, {}),
    (Keypoint, 112, {"r_data": 0, "R": 40, "xd": 72, "sd": 96, "o": 104, "s": 108}),
    (Slab, 24, {}),
    (Keypoint_store, 48, {}),
    (Hist, 48, {}),
    (SIFT3D_Descriptor, 3104, {"xd": 3072}),
    (SIFT3D_Descriptor_store, 32, {}),
    (Sep_FIR_filter, 32, {}),
    (Gauss_filter, 40, {}),
    (GSS_filters, 56, {}),
    (Pyramid, 48, {}),
    (Tri, 48, {}),
    (Mesh, 16, {}),
    (Ransac, 16, {"num_iter": 8}),
    (Tform, 16, {"vtable": 8}),
    (Affine, 48, {"A": 16}),
    (Reg_SIFT3D, 512, {"ref_units": 24, "sift3d": 48, "ran": 352, "desc_src": 368, "desc_ref": 400, "match_src": 432,
                       "match_ref": 464, "nn_thresh": 496, "verbose": 504}),
    (SIFT3D, 304, {"gss": 16, "gpyr": 80, "dog": 128, "im": 176, "peak_thresh": 280,
                   "dense_rotate": 296}),
]


class Sift3dLib:
    """A loaded library that speaks the SIFT3D C API (product or reference oracle).

    ``imutil`` may be a second CDLL when the L1 functions live in a separate shared object (the
    reference splits libimutil / libsift3D; the product exports both layers from one library).
    """

    def __init__(self, sift: C.CDLL, imutil: C.CDLL | None = None, name: str = "?", reg: C.CDLL | None = None):
        self.sift = sift
        self.imutil = imutil if imutil is not None else sift
        self.reg = reg if reg is not None else sift          # registration layer (libreg in the reference)
        self.name = name
        s, u = self.sift, self.imutil
        P = C.POINTER
        u.init_im.argtypes = [P(Image)]
        u.init_im.restype = None
        u.im_free.argtypes = [P(Image)]
        u.im_free.restype = None
        u.im_resize.argtypes = [P(Image)]
        u.im_default_stride.argtypes = [P(Image)]
        u.im_default_stride.restype = None
        u.init_Gauss_filter.argtypes = [P(Gauss_filter), C.c_double, C.c_int]
        u.init_Gauss_incremental_filter.argtypes = [P(Gauss_filter), C.c_double, C.c_double, C.c_int]
        u.cleanup_Gauss_filter.argtypes = [P(Gauss_filter)]
        u.cleanup_Gauss_filter.restype = None
        u.apply_Sep_FIR_filter.argtypes = [P(Image), P(Image), P(Sep_FIR_filter), C.c_double]
        s.init_SIFT3D.argtypes = [P(SIFT3D)]
        s.cleanup_SIFT3D.argtypes = [P(SIFT3D)]
        s.cleanup_SIFT3D.restype = None
        for f in ("set_peak_thresh_SIFT3D", "set_corner_thresh_SIFT3D", "set_sigma_n_SIFT3D",
                  "set_sigma0_SIFT3D"):
            getattr(s, f).argtypes = [P(SIFT3D), C.c_double]
        s.set_num_kp_levels_SIFT3D.argtypes = [P(SIFT3D), C.c_uint]
        s.init_Keypoint_store.argtypes = [P(Keypoint_store)]
        s.init_Keypoint_store.restype = None
        s.cleanup_Keypoint_store.argtypes = [P(Keypoint_store)]
        s.cleanup_Keypoint_store.restype = None
        s.resize_Keypoint_store.argtypes = [P(Keypoint_store), C.c_size_t]
        s.init_SIFT3D_Descriptor_store.argtypes = [P(SIFT3D_Descriptor_store)]
        s.init_SIFT3D_Descriptor_store.restype = None
        s.cleanup_SIFT3D_Descriptor_store.argtypes = [P(SIFT3D_Descriptor_store)]
        s.cleanup_SIFT3D_Descriptor_store.restype = None
        s.SIFT3D_detect_keypoints.argtypes = [P(SIFT3D), P(Image), P(Keypoint_store)]
        s.SIFT3D_extract_descriptors.argtypes = [P(SIFT3D), P(Keypoint_store),
                                                 P(SIFT3D_Descriptor_store)]
        s.SIFT3D_extract_raw_descriptors.argtypes = [P(SIFT3D), P(Image), P(Keypoint_store),
                                                     P(SIFT3D_Descriptor_store)]
        s.SIFT3D_extract_dense_descriptors.argtypes = [P(SIFT3D), P(Image), P(Image)]
        s.SIFT3D_assign_orientations.argtypes = [P(SIFT3D), P(Image), P(Keypoint_store),
                                                 P(P(C.c_double))]
        s.SIFT3D_have_gpyr.argtypes = [P(SIFT3D)]

    # -- helpers that only marshal data ------------------------------------------------------
    def image_from_numpy(self, vol: np.ndarray, units=(1.0, 1.0, 1.0)) -> Image:
        """Make an ``Image`` owning a malloc'd copy of ``vol`` (shape [nz, ny, nx] or
        [nz, ny, nx, nc], float32; x fastest like ``im_default_stride``, imutil.c:1453)."""
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        if vol.ndim == 3:
            vol = vol[..., None]
        nz, ny, nx, nc = vol.shape
        im = Image()
        self.imutil.init_im(C.byref(im))
        im.nx, im.ny, im.nz, im.nc = nx, ny, nz, nc
        im.ux, im.uy, im.uz = units
        self.imutil.im_default_stride(C.byref(im))
        if self.imutil.im_resize(C.byref(im)) != 0:
            raise RuntimeError("im_resize failed")
        C.memmove(im.data, vol.ctypes.data, vol.nbytes)
        return im

    def image_to_numpy(self, im: Image) -> np.ndarray:
        n = im.nx * im.ny * im.nz * im.nc
        a = np.ctypeslib.as_array(im.data, shape=(n,)).copy()
        a = a.reshape(im.nz, im.ny, im.nx, im.nc)
        return a[..., 0] if im.nc == 1 else a

    def free_image(self, im: Image) -> None:
        self.imutil.im_free(C.byref(im))
        im.data = None

    @staticmethod
    def descriptor_store_from_numpy(bins: np.ndarray, xyzs: np.ndarray | None = None):
        """A SIFT3D_Descriptor_store over a numpy-owned buffer (returned too: keep it alive)."""
        k = bins.shape[0]
        raw = np.zeros((k, C.sizeof(SIFT3D_Descriptor)), np.uint8)
        if k:
            raw[:, :DESC_NUMEL * 4] = np.ascontiguousarray(bins, np.float32).view(np.uint8).reshape(k, -1)
        if xyzs is not None and k:
            raw[:, DESC_NUMEL * 4:] = np.ascontiguousarray(xyzs, np.float64).view(np.uint8).reshape(k, -1)
        st = SIFT3D_Descriptor_store()
        st.buf = C.cast(raw.ctypes.data, C.POINTER(SIFT3D_Descriptor))
        st.num = k
        return st, raw

    @staticmethod
    def keypoints_to_numpy(kp: Keypoint_store):
        """Return (coords int64 [K,5] = x,y,z,o,s ; sd float64 [K] ; R float32 [K,3,3])."""
        k = int(kp.slab.num)
        xyzos = np.zeros((k, 5), dtype=np.int64)
        xyz_d = np.zeros((k, 3), dtype=np.float64)
        sd = np.zeros(k, dtype=np.float64)
        R = np.zeros((k, 3, 3), dtype=np.float32)
        for i in range(k):
            key = kp.buf[i]
            xyz_d[i] = (key.xd, key.yd, key.zd)
            xyzos[i] = (int(key.xd), int(key.yd), int(key.zd), key.o, key.s)
            sd[i] = key.sd
            R[i] = np.array(key.r_data[:], dtype=np.float32).reshape(3, 3)
        assert np.all(xyz_d == xyzos[:, :3]), "keypoint coordinates are integers stored as double"
        return xyzos, sd, R

    @staticmethod
    def descriptors_to_numpy(desc: SIFT3D_Descriptor_store):
        """Return (bins float32 [K,768] in memory order 12*(cx+4cy+16cz)+v ; xyzs float64 [K,4])."""
        k = int(desc.num)
        raw = np.ctypeslib.as_array(C.cast(desc.buf, C.POINTER(C.c_uint8)),
                                    shape=(k, C.sizeof(SIFT3D_Descriptor))).copy()
        bins = raw[:, :DESC_NUMEL * 4].copy().view(np.float32).reshape(k, DESC_NUMEL)
        xyzs = raw[:, DESC_NUMEL * 4:].copy().view(np.float64).reshape(k, 4)
        return bins, xyzs
