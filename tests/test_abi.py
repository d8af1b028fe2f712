"""CPU: ABI contract of the drop-in library -- struct layouts equal the reference's and
the shared object loads and exports every symbol the public headers declare.  No compute calls."""
import ctypes as C
import os
import re

import pytest

import sift3d_amd
from sift3d_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_layout_matches_reference():
    for struct, size, offsets in abi.ABI_LAYOUT:
        assert C.sizeof(struct) == size, struct.__name__
        for field, off in offsets.items():
            assert getattr(struct, field).offset == off, (struct.__name__, field)


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    txt = re.sub(r"\(\s*\*\s*\w+\s*\)\s*\([^;{}]*\)\s*;", ";", txt)      # function-pointer members of structs
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", txt)
    return sorted(set(n for n in names if n not in ("_Static_assert", "S3D_ABI_SIZE", "S3D_ABI_OFF", "sizeof")))


@pytest.mark.parametrize("header", ["sift3d_amd.h", "s3d_device.h"])
def test_library_exports_every_declared_symbol(header):
    from sift3d_amd import build as _b
    _b.build()
    lib = sift3d_amd.cdll()
    names = _declared_functions(header)
    assert len(names) > 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"{header}: not exported: {missing}"


def test_reference_headers_static_asserts_compile(tmp_path):
    """The _Static_asserts in include/sift3d_amd.h are evaluated by a C compiler."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "sift3d_amd.h"\nint main(void){return 0;}\n')
    subprocess.run(["gcc", "-std=gnu11", f"-I{ROOT}/include", "-c", str(src), "-o", str(tmp_path / "abi.o")], check=True)


def test_host_only_lifecycle_without_gpu():
    """init/cleanup, setters' range checks and the stores work with no device at all."""
    lib = sift3d_amd.load()
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    assert s.peak_thresh == 0.1 and s.corner_thresh == 0.4 and s.gpyr.num_kp_levels == 3
    assert s.gpyr.sigma_n == 1.15 and s.gpyr.sigma0 == 1.6 and s.gpyr.first_level == -1
    assert lib.sift.set_peak_thresh_SIFT3D(C.byref(s), 0.0) != 0
    assert lib.sift.set_peak_thresh_SIFT3D(C.byref(s), 1.5) != 0
    assert lib.sift.set_corner_thresh_SIFT3D(C.byref(s), -0.1) != 0
    assert lib.sift.set_sigma_n_SIFT3D(C.byref(s), -1.0) != 0
    assert lib.sift.set_peak_thresh_SIFT3D(C.byref(s), 0.2) == 0 and s.peak_thresh == 0.2
    assert lib.sift.SIFT3D_have_gpyr(C.byref(s)) == 0
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    assert lib.sift.resize_Keypoint_store(C.byref(kp), 3) == 0
    assert kp.slab.num == 3 and kp.slab.buf_size == 500 * C.sizeof(abi.Keypoint)
    assert C.addressof(kp.buf[1].r_data) == kp.buf[1].R.data      # R aliases r_data after (re)allocation
    assert lib.sift.resize_Keypoint_store(C.byref(kp), 501) == 0
    assert kp.slab.buf_size == 1000 * C.sizeof(abi.Keypoint)
    # descriptors without keypoints / pyramid are an error, as in the reference
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    lib.sift.resize_Keypoint_store(C.byref(kp), 0)
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) != 0
    lib.sift.cleanup_Keypoint_store(C.byref(kp))
    lib.sift.cleanup_SIFT3D(C.byref(s))


def test_gauss_filter_bank_is_the_references(oracle):
    lib = sift3d_amd.load()
    import numpy as np
    from tests.util import nbitdiff
    for sigma in (0.0, 0.2, 0.538701, 1.1124, 2.45255, 2.8284, 5.0):
        g = abi.Gauss_filter()
        assert lib.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
        t = np.ctypeslib.as_array(g.f.kernel, shape=(g.f.width,)).copy()
        assert nbitdiff(t, oracle.gauss_taps(sigma)) == 0
        lib.imutil.cleanup_Gauss_filter(C.byref(g))
    g = abi.Gauss_filter()
    assert lib.imutil.init_Gauss_incremental_filter(C.byref(g), 2.0, 1.0, 3) != 0   # s_cur > s_next


def test_exported_constants_match_the_reference(reference):
    """libsift3D exports its 19 parameter constants as data symbols (sift.c:34-58); same names, types and values."""
    import ctypes as C
    from sift3d_amd import build as _b
    mine = C.CDLL(_b.build())
    for name in ("peak_thresh_default", "corner_thresh_default", "sigma_n_default", "sigma0_default", "max_eig_ratio",
                 "ori_grad_thresh", "bary_eps", "ori_sig_fctr", "ori_rad_fctr", "desc_sig_fctr", "desc_rad_fctr",
                 "trunc_thresh", "gr"):
        assert C.c_double.in_dll(mine, name).value == C.c_double.in_dll(reference.sift, name).value, name
    assert C.c_int.in_dll(mine, "num_kp_levels_default").value == C.c_int.in_dll(reference.sift, "num_kp_levels_default").value
    for name in ("opt_peak_thresh", "opt_corner_thresh", "opt_num_kp_levels", "opt_sigma_n", "opt_sigma0"):
        a = C.string_at(C.addressof(C.c_char.in_dll(mine, name)))
        b = C.string_at(C.addressof(C.c_char.in_dll(reference.sift, name)))
        assert a == b == name[4:].encode(), name
