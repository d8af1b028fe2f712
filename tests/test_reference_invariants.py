"""GPU: the three invariants of the reference's own suite that earlier rounds had no counterpart for
(wrappers/matlab/Sift3DTest.m): regAnisoTest (:332-358, reg/reg.c:366-429), rawDescriptorTest (:179-201),
rawOrientationTest (:205-242) -- on synthetic volumes, through the C API and the regSift3D program, and, for the resampling
registration, side by side with the unmodified reference (oracle/_ref) with the RANSAC seed pinned."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from sift3d_amd import abi, build as _b, synth
from tests import parity
from tests.test_cli import _csv, BIN, run
from tests.test_host_io import nifti1_bytes
from tests.test_reg import _affine, _bind, _mat_np, libc

pytestmark = pytest.mark.gpu
P = C.POINTER


def _aniso_pair(n=96, nblobs=1500, seed=4):
    """a volume and its every-other-slice copy, units (1, 1, 1) and (1, 1, 2)  (Sift3DTest.m:341-343)"""
    vol = synth.blobs(n, n, n, nblobs, seed)
    return vol, np.ascontiguousarray(vol[::2]), (1.0, 1.0, 1.0), (1.0, 1.0, 2.0)


def _register_resample(L, vol, half, u1, u2):
    _bind(L)
    L.reg.register_SIFT3D_resample.argtypes = [P(abi.Reg_SIFT3D), P(abi.Image), P(abi.Image), C.c_int, C.c_void_p]
    reg = abi.Reg_SIFT3D()
    assert L.reg.init_Reg_SIFT3D(C.byref(reg)) == 0
    src, ref = L.image_from_numpy(vol, u1), L.image_from_numpy(half, u2)
    t = _affine(L)
    libc.srand(1)
    rc = L.reg.register_SIFT3D_resample(C.byref(reg), C.byref(src), C.byref(ref), 0, C.byref(t))      # LINEAR
    out = (rc, _mat_np(t.A) if rc == 0 else None, _mat_np(reg.match_src), _mat_np(reg.match_ref))
    L.reg.cleanup_Reg_SIFT3D(C.byref(reg))
    L.free_image(src)
    L.free_image(ref)
    return out


def _check_aniso_transform(A):
    """Sift3DTest.m:349-357: the linear part within 5e-2 of diag(1, 1, 2), the translation within 5 voxels"""
    want = np.diag([1.0, 1.0, 2.0])
    assert np.abs(A[:, :3] - want).max() <= 5e-2, A
    assert np.abs(A[:, 3]).max() <= 5.0, A


def test_register_resample_aniso(hip, reference, capfd):
    """register_SIFT3D_resample on a volume and its every-other-slice copy: A ~ diag(1, 1, 2) (the reference suite's bound), and --
    same libc rand() seed -- the same matches and transform as the unmodified reference finds (its descriptors differ from ours
    inside 1e-4: the match decisions and hence the inlier sets are the same)."""
    vol, half, u1, u2 = _aniso_pair()
    rc, A, ms, mr = _register_resample(hip, vol, half, u1, u2)
    assert rc == 0 and ms.shape[0] >= 30
    _check_aniso_transform(A)
    rc2, A2, ms2, mr2 = _register_resample(reference, vol, half, u1, u2)
    capfd.readouterr()
    assert rc2 == 0
    _check_aniso_transform(A2)
    assert np.array_equal(ms, ms2) and np.array_equal(mr, mr2)
    assert np.abs(A - A2).max() <= 1e-6


def test_regSift3D_resample_end_to_end(tmp_path):
    """The program with --resample (cli/regSift3D.c:222-242 upstream) on NIfTI files of the same pair."""
    _b.build()
    vol, half, u1, u2 = _aniso_pair()
    for name, v, u in (("src", vol, u1), ("ref", half, u2)):
        with gzip.open(str(tmp_path / f"{name}.nii.gz"), "wb") as f:
            f.write(nifti1_bytes(np.ascontiguousarray(v.transpose(2, 1, 0)), u))
    tf, mt = str(tmp_path / "tform.csv"), str(tmp_path / "matches.csv")
    r = run(os.path.join(BIN, "regSift3D"), "--resample", "--transform", tf, "--matches", mt, str(tmp_path / "src.nii.gz"),
            str(tmp_path / "ref.nii.gz"))
    assert r.returncode == 0, r.stderr
    A = np.array(_csv(tf), np.float64)
    assert A.shape == (3, 4)
    _check_aniso_transform(A)
    assert len(_csv(mt)) >= 30


def _detect(lib, dims=(96, 96, 96), units=(1.0, 1.0, 1.0), nblobs=1500, seed=4):
    vol = synth.blobs(*dims, nblobs, seed)
    s, im, kp = parity.run_detect(lib, vol, units)
    assert int(kp.slab.num) >= 100
    return vol, s, im, kp


def test_raw_descriptors_close_to_pyramid_descriptors(hip):
    """rawDescriptorTest (Sift3DTest.m:179-201): descriptors extracted from the raw image at the keypoints' base-octave
    coordinates lie within 0.2 (absolute, per bin) of those extracted from the scale-space pyramid; the coordinates agree."""
    _, s, im, kp = _detect(hip)
    xyzos, _, _ = hip.keypoints_to_numpy(kp)
    dp, dr = abi.SIFT3D_Descriptor_store(), abi.SIFT3D_Descriptor_store()
    hip.sift.init_SIFT3D_Descriptor_store(C.byref(dp))
    hip.sift.init_SIFT3D_Descriptor_store(C.byref(dr))
    assert hip.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(dp)) == 0
    assert hip.sift.SIFT3D_extract_raw_descriptors(C.byref(s), C.byref(im), C.byref(kp), C.byref(dr)) == 0
    bp, xp = hip.descriptors_to_numpy(dp)
    br, xr = hip.descriptors_to_numpy(dr)
    assert bp.shape == br.shape and bp.shape[0] == len(xyzos)
    assert np.allclose(xp[:, :3], xr[:, :3], rtol=0, atol=1e-9)            # assertElementsAlmostEqual(coordsPyr, coordsRaw)
    assert np.abs(bp - br).max() <= 0.2
    assert np.abs(bp - br).mean() < 0.01                                   # (and they are close on the whole, not just bounded)
    hip.sift.cleanup_SIFT3D_Descriptor_store(C.byref(dp))
    hip.sift.cleanup_SIFT3D_Descriptor_store(C.byref(dr))
    hip.sift.cleanup_SIFT3D(C.byref(s))


def test_raw_orientations_close_to_pyramid_orientations(hip):
    """rawOrientationTest (Sift3DTest.m:205-242): orientations assigned on the raw image against those assigned in the pyramid:
    the median angle between the rotated first basis vectors stays below pi / 8."""
    _, s, im, kp = _detect(hip)
    _, _, R0 = hip.keypoints_to_numpy(kp)
    conf = P(C.c_double)()
    assert hip.sift.SIFT3D_assign_orientations(C.byref(s), C.byref(im), C.byref(kp), C.byref(conf)) == 0
    _, _, R1 = hip.keypoints_to_numpy(kp)
    assert R0.shape == R1.shape
    for col in (True, False):                                              # ori * u: the first column (and, transposed storage, row)
        a = R0[:, :, 0] if col else R0[:, 0, :]
        b = R1[:, :, 0] if col else R1[:, 0, :]
        ang = np.arccos(np.clip(np.abs((a.astype(np.float64) * b).sum(1)), 0.0, 1.0))
        assert np.median(ang) < np.pi / 8, np.median(ang)
    hip.sift.cleanup_SIFT3D(C.byref(s))
