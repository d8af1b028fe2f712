"""Source-level drop-in check: the reference's own example programs (examples/featuresC.c, registerC.c, ioC.c) and its
command-line programs (cli/kpSift3D.c, denseSift3D.c, regSift3D.c)
are compiled UNCHANGED, from where they lie under /root/reference, against include/compat/*.h and
libsift3d_amd.so, and run (device work on the SIMT emulator build, interposed with LD_PRELOAD) on synthetic
volumes placed under the file names the examples hard-code.  Skipped where the reference tree is absent
(the GPU box); nothing of the reference is copied into this repository.
"""
import gzip
import os
import subprocess

import numpy as np
import pytest

from sift3d_amd import build as _b, synth
from tests.test_cli import _csv, _dense_end_to_end, _kp_end_to_end, _nii_f32
from tests.test_host_io import nifti1_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = "/root/reference/examples"
REF_CLI = "/root/reference/cli"
EMU = os.path.join(ROOT, "tests", "emu")

pytestmark = pytest.mark.skipif(not os.path.isdir(EXAMPLES), reason="reference tree not present")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    _b.build()
    subprocess.run(["sh", os.path.join(EMU, "build_emu.sh")], check=True, capture_output=True)
    out = tmp_path_factory.mktemp("examples")
    exes = {}
    for name in ("featuresC", "registerC", "ioC"):
        exe = str(out / name)
        subprocess.run(["gcc", "-std=gnu11", f"-I{ROOT}/include/compat", os.path.join(EXAMPLES, name + ".c"), "-o", exe,
                        f"-L{ROOT}/sift3d_amd/lib", "-lsift3d_amd", "-lm", f"-Wl,-rpath,{ROOT}/sift3d_amd/lib"],
                       check=True, capture_output=True)
        exes[name] = exe
    return exes


def _run(exe, cwd):
    return subprocess.run([exe], cwd=cwd, capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, LD_PRELOAD=os.path.join(EMU, "libsift3d_emu.so")))


def _volumes(cwd):
    nx, ny, nz = 48, 44, 40
    a = synth.blobs(nx, ny, nz, 500, 21)
    b = np.roll(a, (2, 1, 1), axis=(2, 1, 0)).copy()
    for name, v in (("1.nii.gz", a), ("2.nii.gz", b)):
        with gzip.open(os.path.join(cwd, name), "wb") as f:
            f.write(nifti1_bytes(np.ascontiguousarray(v.transpose(2, 1, 0)), (1.0, 1.0, 1.5)))
    return a, b


def test_featuresC(built, oracle, tmp_path):
    a, _ = _volumes(str(tmp_path))
    r = _run(built["featuresC"], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    xyzos, sd, R = oracle.detect(a, (1.0, 1.0, 1.5))
    rows = _csv(str(tmp_path / "1_keys.csv.gz"))
    assert len(rows) == len(xyzos) >= 10
    for row, c, s_ in zip(rows, xyzos, sd):
        assert row[:5] == ["%f" % float(c[0]), "%f" % float(c[1]), "%f" % float(c[2]), "%f" % float(c[3]), "%f" % s_]
    assert len(_csv(str(tmp_path / "1_desc.csv.gz"))) == len(xyzos)
    pts, _ = _nii_f32(str(tmp_path / "1_keys.nii.gz"))
    assert pts.shape == (48, 44, 40) and pts.sum() > 0


def test_ioC(built, tmp_path):
    """ioC reads 1.nii.gz and writes a DICOM series: the read succeeds, the write stops exactly like a reference
    built without DCMTK (SIFT3D_WRAPPER_NOT_COMPILED) -- DICOM is out of scope here."""
    _volumes(str(tmp_path))
    r = _run(built["ioC"], str(tmp_path))
    assert r.returncode == 1 and "DICOM" in r.stderr and "failed to find file" not in r.stderr


def test_registerC(built, tmp_path):
    _volumes(str(tmp_path))
    r = _run(built["registerC"], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    A = np.array(_csv(str(tmp_path / "1_2_affine.csv.gz")), np.float64)
    assert A.shape == (3, 4) and np.abs(A[:, :3] - np.eye(3)).max() < 0.1
    w, _ = _nii_f32(str(tmp_path / "2_warped.nii.gz"))
    assert w.shape == (48, 44, 40) and np.isfinite(w).all() and np.abs(w).max() > 0


@pytest.fixture(scope="module")
def ref_cli(tmp_path_factory):
    """cli/*.c of the reference, as they lie, against include/compat + libsift3d_amd.so (row f3/f4: "the reference's
    command lines").  regSift3D links the three exported defaults (cli/regSift3D.c:83-84)."""
    _b.build()
    subprocess.run(["sh", os.path.join(EMU, "build_emu.sh")], check=True, capture_output=True)
    out = tmp_path_factory.mktemp("refcli")
    exes = {}
    for name in ("kpSift3D", "denseSift3D", "regSift3D"):
        exe = str(out / name)
        subprocess.run(["gcc", "-std=gnu11", "-w", f"-I{ROOT}/include/compat", os.path.join(REF_CLI, name + ".c"), "-o", exe,
                        f"-L{ROOT}/sift3d_amd/lib", "-lsift3d_amd", "-lm", f"-Wl,-rpath,{ROOT}/sift3d_amd/lib"],
                       check=True, capture_output=True)
        exes[name] = exe
    return exes


def test_reference_regSift3D_help_prints_the_exported_defaults(ref_cli):
    r = subprocess.run([ref_cli["regSift3D"], "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("Usage: regSift3D [source.nii] [reference.nii]")
    assert "(default: 0.80)" in r.stdout and "(default: 5.0)" in r.stdout and "(default: 500)" in r.stdout
    # ... and from "Other options:" on it is the text our own cli/regSift3D.c prints (above that line ours leaves out the
    # DICOM output formats, which this library does not write)
    mine = subprocess.run([os.path.join(ROOT, "sift3d_amd", "bin", "regSift3D"), "--help"], capture_output=True, text=True, timeout=60)
    assert mine.stdout.split("Other options:")[1] == r.stdout.split("Other options:")[1]


def test_reference_kpSift3D_source_end_to_end(ref_cli, oracle, tmp_path):
    _kp_end_to_end(ref_cli, oracle, tmp_path, (24, 20, 18), 40, 3, 1, {"LD_PRELOAD": os.path.join(EMU, "libsift3d_emu.so")})


def test_reference_denseSift3D_source_end_to_end(ref_cli, oracle, tmp_path):
    _dense_end_to_end(ref_cli, oracle, tmp_path, (14, 13, 12), {"LD_PRELOAD": os.path.join(EMU, "libsift3d_emu.so")})


def test_reference_regSift3D_source_one_registration(ref_cli, tmp_path):
    from tests.test_reg import _reg_end_to_end
    n = _reg_end_to_end(tmp_path, (48, 44, 40), 500, (2, 1, 1), {"LD_PRELOAD": os.path.join(EMU, "libsift3d_emu.so")},
                        exe=ref_cli["regSift3D"])
    assert n >= 5
