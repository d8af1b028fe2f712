"""The element-wise image helpers of the pyramid build (im_max_abs, im_scale, im_subtract, im_downsample_2x) and
copy_SIFT3D as public entry points: same results as the unmodified reference (oracle/_ref), bit for bit.  CPU: kernels
on the SIMT emulator; GPU: the product library."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sift3d_amd import abi, synth
from sift3d_amd.device import bind_extensions
from tests import parity
from tests.util import nbitdiff

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["sh", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    L = C.CDLL(os.path.join(EMU_DIR, "libsift3d_emu.so"))
    lib = abi.Sift3dLib(L, None, "emulated")
    bind_extensions(L)
    return lib


def _bind(lib):
    IP = C.POINTER(abi.Image)
    lib.imutil.im_max_abs.restype, lib.imutil.im_max_abs.argtypes = C.c_float, [IP]
    lib.imutil.im_scale.restype, lib.imutil.im_scale.argtypes = None, [IP]
    lib.imutil.im_subtract.restype, lib.imutil.im_subtract.argtypes = C.c_int, [IP, IP, IP]
    lib.imutil.im_downsample_2x.restype, lib.imutil.im_downsample_2x.argtypes = C.c_int, [IP, IP]
    lib.sift.copy_SIFT3D.restype = C.c_int
    lib.sift.copy_SIFT3D.argtypes = [C.POINTER(abi.SIFT3D), C.POINTER(abi.SIFT3D)]


def _vol(shape, seed):
    return (np.random.default_rng(seed).standard_normal(shape) * 37.0).astype(np.float32)


def check_image_ops(lib, ref):
    _bind(lib)
    _bind(ref)
    for shape in ((9, 11, 13), (8, 12, 16, 3), (7, 5, 6, 12)):
        a, b = _vol(shape, 1), _vol(shape, 2)
        res = []
        for l in (lib, ref):
            ia, ib = l.image_from_numpy(a, (1, 1.5, 2)), l.image_from_numpy(b)
            m = l.imutil.im_max_abs(C.byref(ia))
            sub, dn = abi.Image(), abi.Image()
            l.imutil.init_im(C.byref(sub))
            l.imutil.init_im(C.byref(dn))
            assert l.imutil.im_subtract(C.byref(ia), C.byref(ib), C.byref(sub)) == 0
            assert l.imutil.im_downsample_2x(C.byref(ia), C.byref(dn)) == 0
            l.imutil.im_scale(C.byref(ib))
            res.append((m, l.image_to_numpy(sub), (sub.ux, sub.uy, sub.uz), l.image_to_numpy(dn), (dn.nx, dn.ny, dn.nz, dn.nc),
                        l.image_to_numpy(ib)))
            for im in (ia, ib, sub, dn):
                l.free_image(im)
        got, want = res
        assert got[0] == want[0] == np.abs(a).max()
        assert nbitdiff(got[1], want[1]) == 0 and got[2] == want[2]
        assert got[4] == want[4] and nbitdiff(got[3], want[3]) == 0
        assert nbitdiff(got[5], want[5]) == 0 and np.abs(got[5]).max() == 1.0
    # zero image: im_scale leaves it alone; mismatching dims: im_subtract fails like the reference
    for l in (lib, ref):
        z = l.image_from_numpy(np.zeros((4, 5, 6), np.float32))
        l.imutil.im_scale(C.byref(z))
        assert l.imutil.im_max_abs(C.byref(z)) == 0.0 and not l.image_to_numpy(z).any()
        o = l.image_from_numpy(np.zeros((4, 5, 7), np.float32))
        d = abi.Image()
        l.imutil.init_im(C.byref(d))
        assert l.imutil.im_subtract(C.byref(z), C.byref(o), C.byref(d)) != 0
        for im in (z, o):
            l.free_image(im)


def check_copy_sift3d(lib):
    """Descriptors extracted through a copy of a detector equal those of the original; parameters travel."""
    _bind(lib)
    vol = synth.blobs(40, 36, 32, 120, 4)
    s, im, kp = parity.run_detect(lib, vol, (1, 1, 1.5), {"peak_thresh": 0.08, "corner_thresh": 0.3})
    s.dense_rotate = 1
    c = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(c)) == 0
    assert lib.sift.copy_SIFT3D(C.byref(s), C.byref(c)) == 0
    assert (c.peak_thresh, c.corner_thresh, c.dense_rotate) == (s.peak_thresh, s.corner_thresh, 1)
    assert (c.gpyr.num_kp_levels, c.gpyr.sigma0, c.gpyr.sigma_n) == (s.gpyr.num_kp_levels, s.gpyr.sigma0, s.gpyr.sigma_n)
    assert (c.im.nx, c.im.ny, c.im.nz, c.im.uz) == (40, 36, 32, 1.5) and c.gpyr.num_octaves == s.gpyr.num_octaves
    out = []
    for det in (s, c):
        d = abi.SIFT3D_Descriptor_store()
        lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
        assert lib.sift.SIFT3D_extract_descriptors(C.byref(det), C.byref(kp), C.byref(d)) == 0
        out.append(lib.descriptors_to_numpy(d)[0])
        lib.sift.cleanup_SIFT3D_Descriptor_store(C.byref(d))
    assert len(out[0]) > 5 and nbitdiff(out[0], out[1]) == 0
    # the copy is independent: a new detect on the original leaves it untouched
    s2, im2, kp2 = parity.run_detect(lib, synth.blobs(40, 36, 32, 120, 9), (1, 1, 1.5))
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(c), C.byref(kp), C.byref(d)) == 0
    assert nbitdiff(lib.descriptors_to_numpy(d)[0], out[0]) == 0
    # an empty detector copies as parameters only
    e, f = abi.SIFT3D(), abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(e)) == 0 and lib.sift.init_SIFT3D(C.byref(f)) == 0
    assert lib.sift.set_peak_thresh_SIFT3D(C.byref(e), 0.2) == 0
    assert lib.sift.copy_SIFT3D(C.byref(e), C.byref(f)) == 0 and f.peak_thresh == 0.2 and f.im.nx == 0
    for det in (s, c, s2, e, f):
        lib.sift.cleanup_SIFT3D(C.byref(det))


def test_image_ops_emulated(emu, reference):
    check_image_ops(emu, reference)


def test_copy_sift3d_emulated(emu):
    check_copy_sift3d(emu)


def test_copy_sift3d_reference_behaves_the_same(reference):
    check_copy_sift3d(reference)


@pytest.mark.gpu
def test_image_ops_gpu(hip, reference):
    check_image_ops(hip, reference)


@pytest.mark.gpu
def test_copy_sift3d_gpu(hip):
    check_copy_sift3d(hip)
