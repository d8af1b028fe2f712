"""The command-line programs built from cli/*.c: same argument surface, messages and
output formats as the reference's kpSift3D / denseSift3D.

CPU part: option handling that never reaches the device.  GPU part (-m gpu): end to end on a NIfTI file
written by the test-side writer of tests/test_host_io.py, outputs compared with the oracle's results
formatted the way the reference's writers format them ("%f", sift.c:3143-3230).
"""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from sift3d_amd import build as _b, synth
from tests.test_host_io import nifti1_bytes
from tests.util import rel_close

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sift3d_amd", "bin")


@pytest.fixture(scope="module")
def progs():
    _b.build()
    return {p: os.path.join(BIN, p) for p in ("kpSift3D", "denseSift3D")}


def run(*argv, cwd=None, env=None):
    return subprocess.run(list(argv), capture_output=True, text=True, cwd=cwd, timeout=600,
                          env=None if env is None else dict(os.environ, **env))


@pytest.fixture(scope="module")
def emu_env():
    """CPU stand-in for the device (test infrastructure): the SIMT-emulator build of the same sources,
    interposed over libsift3d_amd.so's symbols, so the programs run end to end without a GPU."""
    emu = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.run(["sh", os.path.join(emu, "build_emu.sh")], check=True, capture_output=True)
    return {"LD_PRELOAD": os.path.join(emu, "libsift3d_emu.so")}


def test_help_version_and_usage_errors(progs, tmp_path):
    r = run(progs["kpSift3D"], "--help")
    assert r.returncode == 0 and r.stdout.startswith("Usage: kpSift3D [image.nii]") and "--peak_thresh [value]" in r.stdout
    assert "(default: 0.10)" in r.stdout and "(default: 1.60)" in r.stdout
    r = run(progs["kpSift3D"], "--version")
    assert r.returncode == 0 and r.stdout.startswith("SIFT3D version 1.4.6")
    r = run(progs["denseSift3D"], "--help")
    assert r.returncode == 0 and r.stdout.startswith("Usage: denseSift3D [input.nii] [descriptors%.nii]")
    for args, msg in ((["x.nii"], "No outputs specified."), (["--keys", "k.csv"], "Not enough arguments."),
                      (["--keys", "k.csv", "a.nii", "b.nii"], "Too many arguments.")):
        r = run(progs["kpSift3D"], *args)
        assert r.returncode == 1 and f"kpSift3D: {msg}" in r.stderr and 'Use "kpSift3D --help"' in r.stderr
    r = run(progs["kpSift3D"], "--keys", "k.csv", str(tmp_path / "missing.nii"))
    assert r.returncode == 1 and "failed to find file" in r.stderr and "Could not read image." in r.stderr
    r = run(progs["kpSift3D"], "--peak_thresh", "3", "--keys", "k.csv", "x.nii")
    assert r.returncode == 1 and "peak_thresh must be in the interval (0, 1]" in r.stderr
    r = run(progs["kpSift3D"], "--bogus", "1", "--keys", "k.csv", "x.nii")
    assert r.returncode == 1
    r = run(progs["denseSift3D"], "in.nii")
    assert r.returncode == 1 and "denseSift3d: Not enough arguments." in r.stderr
    r = run(progs["denseSift3D"], "a", "b", "c")
    assert r.returncode == 1 and "Too many arguments." in r.stderr
    vol = np.zeros((8, 8, 8), np.float32)
    (tmp_path / "v.nii").write_bytes(nifti1_bytes(vol, (1, 1, 1)))
    r = run(progs["denseSift3D"], str(tmp_path / "v.nii"), str(tmp_path / "out.nii"))
    assert r.returncode == 1 and "output filename must contain '%'." in r.stderr


def _csv(path):
    with (gzip.open(path, "rt") if path.endswith(".gz") else open(path, "rt")) as f:
        return [line.split(",") for line in f.read().splitlines()]


def _nii_f32(path):
    raw = gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()
    dim = struct.unpack_from("<8h", raw, 40)
    assert raw[344:348] == b"n+1\0" and struct.unpack_from("<h", raw, 70)[0] == 16
    shape = dim[1:1 + dim[0]]
    return np.frombuffer(raw, "<f4", offset=352).reshape(shape, order="F"), struct.unpack_from("<8f", raw, 76)[1:4]


def test_kpSift3D_emulated(progs, oracle, tmp_path, emu_env):
    _kp_end_to_end(progs, oracle, tmp_path, (24, 20, 18), 40, 3, 1, emu_env)


@pytest.mark.gpu
def test_kpSift3D_end_to_end(progs, oracle, tmp_path):
    _kp_end_to_end(progs, oracle, tmp_path, (128, 128, 128), 2600, 11, 200, None)      # the size of the reference's own example volumes


def _kp_end_to_end(progs, oracle, tmp_path, dims, nblobs, seed, min_kp, env):
    nx, ny, nz = dims
    units = (1.0, 1.0, 1.5)
    vol = synth.blobs(nx, ny, nz, nblobs, seed)                             # [z, y, x]
    (tmp_path / "in").mkdir()
    src = str(tmp_path / "in" / "vol.nii.gz")
    with gzip.open(src, "wb") as f:
        f.write(nifti1_bytes(np.ascontiguousarray(vol.transpose(2, 1, 0)), units))
    keys, desc, draw = (str(tmp_path / "out" / n) for n in ("keys.csv", "desc.csv.gz", "points.nii.gz"))
    r = run(progs["kpSift3D"], "--peak_thresh", "0.08", "--keys", keys, "--desc", desc, "--draw", draw, src, env=env)
    assert r.returncode == 0, r.stderr

    oracle.set_params(peak=0.08)
    try:
        xyzos, sd, R = oracle.detect(vol, units)
        wb, wx = oracle.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
    finally:
        oracle.set_params()
    assert len(xyzos) >= min_kp
    rows = _csv(keys)
    assert len(rows) == len(xyzos) and all(len(r_) == 14 for r_ in rows)
    for row, c, s_, Rk in zip(rows, xyzos, sd, R):
        assert row[:5] == ["%f" % float(c[0]), "%f" % float(c[1]), "%f" % float(c[2]), "%f" % float(c[3]), "%f" % s_]
        assert np.abs(np.array(row[5:], np.float64) - Rk.ravel().astype(np.float64)).max() <= 1e-5 + 1e-6
    drows = _csv(desc)
    assert len(drows) == len(xyzos) and all(len(r_) == 771 for r_ in drows)
    got = np.array(drows, np.float64)
    assert np.array_equal(got[:, :3], np.array([["%f" % np.float32(v) for v in row[:3]] for row in wx], np.float64))
    # bins: 1e-4 relative (the device tolerance) on top of the six printed decimals
    assert rel_close(got[:, 3:], wb.astype(np.float64), rtol=1e-4, atol=1e-6).all()
    pts, pu = _nii_f32(draw)
    assert pts.shape == (nx, ny, nz) and pu == (1.0, 1.0, 1.0)              # draw_points output carries default units
    want = np.zeros((nx, ny, nz), np.float32)
    for c in xyzos:
        p = (c[:3] * 2 ** c[3]).astype(int)
        lo, hi = np.maximum(p - 1, 0), np.minimum(p + 1, np.array([nx, ny, nz]) - 1)
        want[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = 1.0
    assert np.array_equal(pts, want)


def test_denseSift3D_emulated(progs, oracle, tmp_path, emu_env):
    _dense_end_to_end(progs, oracle, tmp_path, (14, 13, 12), emu_env)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,units", [((22, 20, 18), (1.0, 1.0, 2.0)),
                                        ((80, 72, 66), (0.8, 0.8, 1.5)),      # anisotropic slices: the generic 12-channel passes
                                        ((96, 64, 64), (1.0, 1.0, 1.0))])     # unit voxels: the fused front end (k_bary_x_wave) + marches
def test_denseSift3D_end_to_end(progs, oracle, tmp_path, dims, units):
    _dense_end_to_end(progs, oracle, tmp_path, dims, None, units)


def _dense_end_to_end(progs, oracle, tmp_path, dims, env, units=(1.0, 1.0, 2.0)):
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, max(30, nx * ny * nz // 2000), 5) * 37.0 + 3.0
    src = str(tmp_path / "vol.nii")
    open(src, "wb").write(nifti1_bytes(np.ascontiguousarray(vol.transpose(2, 1, 0)), units))
    r = run(progs["denseSift3D"], src, str(tmp_path / "d" / "bin%.nii.gz"), env=env)
    assert r.returncode == 0, r.stderr
    want = oracle.dense(vol, units)                                         # [z, y, x, 12]
    for c in range(12):
        a, u = _nii_f32(str(tmp_path / "d" / f"bin{c}.nii.gz"))
        assert a.shape == (nx, ny, nz)
        assert np.array_equal(a.transpose(2, 1, 0), want[..., c]), c        # dense path is bit exact
    assert not os.path.exists(str(tmp_path / "d" / "bin12.nii.gz"))
