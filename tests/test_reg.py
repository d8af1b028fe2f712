"""Registration tail: RANSAC affine, the inverse warp, Reg_SIFT3D, regSift3D.

CPU (reference build under oracle/_ref, incl. libreg.so compiled from reg/reg.c): the host-side estimator on
identical inputs with the same libc rand() seed; the device warp through the SIMT emulator.  GPU (-m gpu): the
warp kernel and the program end to end.  The reference itself only promises 5e-2 agreement for registration
(Sift3DTest.m:319-324: unseeded RANSAC); with the seed pinned the two estimators agree far tighter, and the
tri-linear warp is bit-identical.
"""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

import sift3d_amd
from sift3d_amd import abi, build as _b, synth
from tests.test_cli import _csv, _nii_f32, BIN, run
from tests.test_host_io import nifti1_bytes
from tests.util import nbitdiff

P = C.POINTER
libc = C.CDLL(None)


@pytest.fixture(scope="module")
def host():
    return sift3d_amd.load()


def _bind(L):
    r, u = L.reg, L.imutil
    u.init_Mat_rm.argtypes = [P(abi.Mat_rm), C.c_int, C.c_int, C.c_int, C.c_int]
    u.cleanup_Mat_rm.argtypes = [P(abi.Mat_rm)]
    u.cleanup_Mat_rm.restype = None
    u.init_tform.argtypes = [C.c_void_p, C.c_int]
    u.cleanup_tform.argtypes = [C.c_void_p]
    u.cleanup_tform.restype = None
    u.init_Ransac.argtypes = [P(abi.Ransac)]
    u.init_Ransac.restype = None
    u.find_tform_ransac.argtypes = [P(abi.Ransac), P(abi.Mat_rm), P(abi.Mat_rm), C.c_void_p]
    u.write_tform.argtypes = [C.c_char_p, C.c_void_p]
    u.im_inv_transform.argtypes = [C.c_void_p, P(abi.Image), C.c_int, C.c_int, P(abi.Image)]
    u.im_resample.argtypes = [P(abi.Image), P(C.c_double), C.c_int, P(abi.Image)]
    u.Affine_set_mat.argtypes = [P(abi.Mat_rm), P(abi.Affine)]
    r.init_Reg_SIFT3D.argtypes = [P(abi.Reg_SIFT3D)]
    r.cleanup_Reg_SIFT3D.argtypes = [P(abi.Reg_SIFT3D)]
    r.cleanup_Reg_SIFT3D.restype = None
    r.register_SIFT3D.argtypes = [P(abi.Reg_SIFT3D), C.c_void_p]
    return L


def _mat(L, a):
    a = np.ascontiguousarray(a, np.float64)
    m = abi.Mat_rm()
    assert L.imutil.init_Mat_rm(C.byref(m), a.shape[0], a.shape[1], 0, 0) == 0
    if a.size:
        C.memmove(m.data, a.ctypes.data, a.nbytes)
    return m


def _mat_np(m):
    if m.num_rows == 0:
        return np.zeros((0, m.num_cols))
    return np.ctypeslib.as_array(C.cast(m.data, P(C.c_double)), (m.num_rows, m.num_cols)).copy()


def _affine(L, A=None):
    t = abi.Affine()
    assert L.imutil.init_tform(C.byref(t), 0) == 0
    if A is not None:
        m = _mat(L, A)
        assert L.imutil.Affine_set_mat(C.byref(m), C.byref(t)) == 0
        L.imutil.cleanup_Mat_rm(C.byref(m))
    return t


def _points(n, outliers, seed, A):
    rng = np.random.default_rng(seed)
    ref = rng.random((n, 3)) * 100
    src = ref @ A[:, :3].T + A[:, 3] + rng.standard_normal((n, 3)) * 0.3
    bad = rng.choice(n, outliers, replace=False)
    src[bad] = rng.random((outliers, 3)) * 100
    return src, ref


A_TRUE = np.array([[1.02, 0.03, -0.01, 4.0], [-0.02, 0.97, 0.05, -3.0], [0.01, -0.04, 1.01, 2.5]])


@pytest.mark.parametrize("n,outliers,seed", [(60, 20, 1), (200, 120, 2), (12, 2, 3), (5, 0, 4)])
def test_find_tform_ransac_matches_reference(host, reference, n, outliers, seed, capfd):
    _bind(host), _bind(reference)
    src, ref = _points(n, outliers, seed, A_TRUE)
    out = []
    for L in (host, reference):
        ran = abi.Ransac()
        L.imutil.init_Ransac(C.byref(ran))
        assert (ran.err_thresh, ran.num_iter) == (5.0, 500)
        ms, mr = _mat(L, src), _mat(L, ref)
        t = _affine(L)
        libc.srand(1)
        rc = L.imutil.find_tform_ransac(C.byref(ran), C.byref(ms), C.byref(mr), C.byref(t))
        out.append((rc, _mat_np(t.A) if rc == 0 else None))
    capfd.readouterr()
    assert out[0][0] == out[1][0] == 0
    # same samples (same rand() stream), same consensus set; the solvers differ only in rounding
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-9
    assert np.abs(out[0][1] - A_TRUE).max() < 1.0


def test_ransac_failures_like_the_reference(host, reference, capfd):
    _bind(host), _bind(reference)
    rng = np.random.default_rng(0)
    for src, ref in ((rng.random((3, 3)), rng.random((3, 3))),                    # fewer points than terms
                     (rng.random((30, 3)) * 100, rng.random((30, 3)) * 100)):      # no consensus
        rcs = []
        for L in (host, reference):
            ran = abi.Ransac()
            L.imutil.init_Ransac(C.byref(ran))
            ran.err_thresh = 0.5
            ms, mr = _mat(L, src), _mat(L, ref)
            t = _affine(L)
            libc.srand(1)
            rcs.append(L.imutil.find_tform_ransac(C.byref(ran), C.byref(ms), C.byref(mr), C.byref(t)))
        assert rcs[0] == rcs[1] != 0
    capfd.readouterr()


def test_register_on_given_descriptors_matches_reference(host, reference, capfd):
    """Reg_SIFT3D with descriptor stores filled in by hand (detection is covered elsewhere): matches are
    integers and must be identical; the transform agrees to rounding.  Needs a device for SIFT3D_nn_match, so the
    product side runs on the emulator build here."""
    from tests.test_emu_parity import EMU_DIR
    from sift3d_amd.device import bind_extensions
    from tests.util import match_sets, rand_desc
    subprocess.run(["sh", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    Lemu = C.CDLL(os.path.join(EMU_DIR, "libsift3d_emu.so"))
    emu = abi.Sift3dLib(Lemu, None, "emulated")
    bind_extensions(Lemu)
    _bind(emu), _bind(reference)
    rng = np.random.default_rng(5)
    d1 = rand_desc(90, 7)
    d2 = match_sets(d1, 8)
    x1 = np.c_[rng.random((90, 3)) * 80, np.full(90, 1.6)]
    # d2 rows 0..89 are a permutation of d1 with noise: recover it to place consistent coordinates
    perm = np.random.default_rng(8).permutation(90)
    x2 = np.c_[rng.random((d2.shape[0], 3)) * 80, np.full(d2.shape[0], 1.6)]
    x2[:90, :3] = (x1[perm, :3] - A_TRUE[:, 3]) @ np.linalg.inv(A_TRUE[:, :3]).T
    out = []
    for L in (emu, reference):
        reg = abi.Reg_SIFT3D()
        assert L.reg.init_Reg_SIFT3D(C.byref(reg)) == 0
        s1, keep1 = abi.Sift3dLib.descriptor_store_from_numpy(d1, x1)
        s2, keep2 = abi.Sift3dLib.descriptor_store_from_numpy(d2, x2)
        reg.desc_src, reg.desc_ref = s1, s2
        for k in range(3):
            reg.src_units[k], reg.ref_units[k] = (1.0, 1.0, 2.0)[k], (1.0, 0.8, 2.0)[k]
        reg.ran.err_thresh = 1.0
        t = _affine(L)
        libc.srand(1)
        rc = L.reg.register_SIFT3D(C.byref(reg), C.byref(t))
        out.append((rc, _mat_np(reg.match_src), _mat_np(reg.match_ref), _mat_np(t.A) if rc == 0 else None))
        reg.desc_src = abi.SIFT3D_Descriptor_store()        # numpy owns those buffers
        reg.desc_ref = abi.SIFT3D_Descriptor_store()
    capfd.readouterr()
    assert out[0][0] == out[1][0] == 0
    assert out[0][1].shape[0] >= 20
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][3] - out[1][3]).max() <= 1e-8


def _warp_case(L, interp, nc, seed=3):
    rng = np.random.default_rng(seed)
    src = rng.standard_normal((14, 16, 18) if nc == 1 else (14, 16, 18, nc)).astype(np.float32)   # [z, y, x(, c)]
    A = np.array([[0.9, 0.1, 0.05, 1.3], [-0.08, 1.05, 0.02, -0.7], [0.03, -0.06, 0.95, 2.2]])
    t = _affine(L, A)
    im = L.image_from_numpy(src, (1, 1, 1))
    dst = L.image_from_numpy(np.zeros((12, 20, 22) if nc == 1 else (12, 20, 22, nc), np.float32), (1, 1, 1))
    assert L.imutil.im_inv_transform(C.byref(t), C.byref(im), interp, 0, C.byref(dst)) == 0
    return L.image_to_numpy(dst)


@pytest.mark.parametrize("interp,nc", [(0, 1), (0, 3), (1, 1)])
def test_inv_transform_emulated_vs_reference(reference, interp, nc):
    from tests.test_emu_parity import EMU_DIR
    from sift3d_amd.device import bind_extensions
    subprocess.run(["sh", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    Lemu = C.CDLL(os.path.join(EMU_DIR, "libsift3d_emu.so"))
    emu = abi.Sift3dLib(Lemu, None, "emulated")
    bind_extensions(Lemu)
    _bind(emu), _bind(reference)
    got, want = _warp_case(emu, interp, nc), _warp_case(reference, interp, nc)
    assert np.abs(want).max() > 0.1 and (want == 0).any()                  # part of the output falls outside
    if interp == 0:
        assert nbitdiff(got, want) == 0
    else:
        assert np.abs(got - want).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("interp,nc", [(0, 1), (0, 3), (1, 1)])
def test_inv_transform_gpu_vs_reference(host, reference, interp, nc):
    _bind(host), _bind(reference)
    got, want = _warp_case(host, interp, nc), _warp_case(reference, interp, nc)
    if interp == 0:
        assert nbitdiff(got, want) == 0
    else:
        assert np.abs(got - want).max() <= 1e-6


@pytest.mark.gpu
def test_resample_gpu_vs_reference(host, reference):
    _bind(host), _bind(reference)
    src = np.random.default_rng(1).standard_normal((10, 12, 14)).astype(np.float32)
    outs = []
    for L in (host, reference):
        im = L.image_from_numpy(src, (1.0, 2.0, 3.0))
        dst = abi.Image()
        L.imutil.init_im(C.byref(dst))
        units = (C.c_double * 3)(1.0, 1.5, 1.0)
        assert L.imutil.im_resample(C.byref(im), units, 0, C.byref(dst)) == 0
        outs.append((L.image_to_numpy(dst), (dst.ux, dst.uy, dst.uz)))
    assert outs[0][0].shape == outs[1][0].shape == (30, 16, 14) and outs[0][1] == outs[1][1] == (1.0, 1.5, 1.0)
    assert nbitdiff(outs[0][0], outs[1][0]) == 0


def _reg_end_to_end(tmp_path, dims, nblobs, shift, env, exe=None):
    _b.build()
    exe = exe or os.path.join(BIN, "regSift3D")
    nx, ny, nz = dims
    a = synth.blobs(nx, ny, nz, nblobs, 21)
    b = np.roll(a, shift, axis=(2, 1, 0)).copy()                          # ref(x) = src(x - shift)
    units = (1.0, 1.0, 1.5)
    for name, v in (("src", a), ("ref", b)):
        with gzip.open(str(tmp_path / f"{name}.nii.gz"), "wb") as f:
            f.write(nifti1_bytes(np.ascontiguousarray(v.transpose(2, 1, 0)), units))
    mt, tf, wp = (str(tmp_path / "out" / n) for n in ("matches.csv", "tform.csv", "warped.nii.gz"))
    cc, ky, ln = (str(tmp_path / "out" / n) for n in ("concat.nii.gz", "keys.nii.gz", "lines.nii.gz"))
    r = run(exe, "--matches", mt, "--transform", tf, "--warped", wp, "--concat", cc, "--keys", ky,
            "--lines", ln, str(tmp_path / "src.nii.gz"), str(tmp_path / "ref.nii.gz"), env=env)
    assert r.returncode == 0, r.stderr
    for path in (cc, ky, ln):                                             # source | reference side by side
        pic, _ = _nii_f32(path)
        assert pic.shape == (2 * nx, ny, nz) and np.isfinite(pic).all()
    assert _nii_f32(ky)[0].sum() > 0 and _nii_f32(ln)[0].sum() > 0
    m = np.array(_csv(mt), np.float64)
    assert m.shape[1] == 6 and m.shape[0] >= 5
    d = m[:, 3:] - m[:, :3]                                                # ref - src coordinates of a match
    # keypoints of octave o are localised to 2^o voxels of the base grid
    good = np.all(np.abs(d - np.array(shift, np.float64)) <= 4.0, axis=1)
    assert good.mean() > 0.7, (good.mean(), d[:10])
    A = np.array(_csv(tf), np.float64)
    assert A.shape == (3, 4)
    # the transform maps reference voxels to source voxels: x_src = x_ref - shift
    # (keypoints sit on integer voxels, so a handful of matches pins the map only to a fraction of a voxel)
    ctr = np.array([nx, ny, nz], np.float64) / 2
    assert np.abs(A[:, :3] - np.eye(3)).max() < 0.1
    assert np.abs(A[:, :3] @ ctr + A[:, 3] - (ctr - np.array(shift))).max() < 1.5
    w, wu = _nii_f32(wp)
    assert w.shape == (nx, ny, nz)
    core = tuple(slice(6, -6) for _ in range(3))
    want = b.transpose(2, 1, 0)[core]
    assert np.abs(w[core] - want).mean() < 0.25 * np.abs(want).mean() + 1e-3   # sub-voxel residual of the estimate
    return m.shape[0]


def test_regSift3D_emulated(tmp_path):
    from tests.test_emu_parity import EMU_DIR
    subprocess.run(["sh", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    _reg_end_to_end(tmp_path, (48, 44, 40), 500, (2, 1, 1), {"LD_PRELOAD": os.path.join(EMU_DIR, "libsift3d_emu.so")})


@pytest.mark.gpu
def test_regSift3D_end_to_end(tmp_path):
    assert _reg_end_to_end(tmp_path, (96, 80, 64), 500, (3, -2, 1), None) >= 20


def test_regSift3D_usage(tmp_path):
    _b.build()
    exe = os.path.join(BIN, "regSift3D")
    r = run(exe, "--help")
    assert r.returncode == 0 and r.stdout.startswith("Usage: regSift3D [source.nii] [reference.nii]")
    assert "(default: 0.80)" in r.stdout and "(default: 5.0)" in r.stdout and "(default: 500)" in r.stdout
    for args, msg in ((["a.nii", "b.nii"], "No outputs were specified."), (["--matches", "m.csv", "a.nii"], "Not enough arguments."),
                      (["--matches", "m.csv", "a", "b", "c"], "Too many arguments."),
                      (["--type", "tps", "--matches", "m.csv", "a", "b"], "Unrecognized transformation type: tps"),
                      (["--nn_thresh", "2", "--matches", "m.csv", "a", "b"], "Invalid value for nn_thresh.")):
        r = run(exe, *args)
        assert r.returncode == 1 and msg in r.stderr, (args, r.stderr)


def test_draw_matches_matches_reference(host, reference):
    """The picture outputs of regSift3D (host only): concatenated volumes, keypoint cubes, match segments --
    voxel for voxel the reference's images."""
    rng = np.random.default_rng(11)
    left = rng.standard_normal((9, 12, 14)).astype(np.float32)            # [z, y, x]
    right = rng.standard_normal((11, 10, 13)).astype(np.float32)
    kl = rng.random((15, 3)) * np.array([14, 12, 9]) * 1.1 - 0.5           # some fall outside
    kr = rng.random((12, 3)) * np.array([13, 10, 11])
    ml = rng.random((10, 3)) * np.array([13.9, 11.9, 8.9])
    mr = rng.random((10, 3)) * np.array([12.9, 9.9, 10.9])
    ml[0] = mr[0] = (3.2, 4.0, 2.0)                                        # a vertical segment after the shift? no: same x + pad
    ml[1, 0] = 20.0                                                        # outside the left image but inside the canvas
    outs = []
    for L in (host, reference):
        L.sift.draw_matches.argtypes = [P(abi.Image)] * 2 + [P(abi.Mat_rm)] * 4 + [P(abi.Image)] * 3
        L.imutil.init_Mat_rm.argtypes = [P(abi.Mat_rm), C.c_int, C.c_int, C.c_int, C.c_int]
        il, ir = L.image_from_numpy(left, (1, 1, 2)), L.image_from_numpy(right, (1, 1, 1))
        mats = [_mat(L, a) for a in (kl, kr, ml, mr)]
        ims = [abi.Image() for _ in range(3)]
        for im in ims:
            L.imutil.init_im(C.byref(im))
        assert L.sift.draw_matches(C.byref(il), C.byref(ir), *[C.byref(m) for m in mats], *[C.byref(im) for im in ims]) == 0
        outs.append([L.image_to_numpy(im) for im in ims])
        assert L.sift.draw_matches(C.byref(il), C.byref(ir), None, None, None, None, None, None, None) != 0
    for a, b in zip(*outs):
        assert a.shape == b.shape == (11, 12, 27) and np.array_equal(a, b)
    assert outs[0][1].sum() > 0 and outs[0][2].sum() > 0
