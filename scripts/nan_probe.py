#!/usr/bin/env python3
"""Non-finite voxels: the reference (oracle/_ref), the restatement (oracle/s3d_oracle.c) and the product's kernel sources
under the SIMT emulator (tests/emu) on the same volumes -- keypoints, orientations and descriptors side by side.
CPU only; test tooling."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from sift3d_amd import abi, synth           # noqa: E402
from sift3d_amd.device import bind_extensions   # noqa: E402
from tests import parity                    # noqa: E402


def emu_lib():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["sh", os.path.join(d, "build_emu.sh")], check=True, capture_output=True)
    L = C.CDLL(os.path.join(d, "libsift3d_emu.so"))
    lib = abi.Sift3dLib(L, None, "emulated")
    bind_extensions(L)
    return lib


def run(lib, vol, units, describe=True):
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    im = lib.image_from_numpy(vol, units)
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    rc = lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp))
    if rc != 0:
        return ("detect failed",)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    bins = None
    if describe and len(xyzos):
        d = abi.SIFT3D_Descriptor_store()
        lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
        rc = lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d))
        if rc != 0:
            return xyzos, sd, R, "describe failed"
        bins, _ = lib.descriptors_to_numpy(d)
    return xyzos, sd, R, bins


def cases(dims, seed):
    nx, ny, nz = dims
    base = synth.blobs(nx, ny, nz, max(30, nx * ny * nz // 800), seed)
    def put(idx, val):
        v = base.copy()
        v[idx] = val
        return v
    yield "nan_first", put((0, 0, 0), np.nan)
    yield "nan_interior", put((7, 6, 5), np.nan)
    yield "nan_center", put((nz // 2, ny // 2, nx // 2), np.nan)
    yield "nan_last", put((nz - 1, ny - 1, nx - 1), np.nan)
    v = base.copy(); v[: nz // 4] = np.nan
    yield "nan_slab_low", v
    v = base.copy(); v[-(nz // 4):] = np.nan
    yield "nan_slab_high", v
    v = base.copy(); v[:, :, : nx // 5] = np.nan
    yield "nan_xband", v
    yield "inf_interior", put((7, 6, 5), np.inf)
    yield "neginf_center", put((nz // 2, ny // 2, nx // 2), -np.inf)
    yield "two_nan", put((3, 3, 3), np.nan) * np.where(np.arange(nx) == nx - 2, np.nan, 1.0).astype(np.float32)[None, None, :] if False else put((nz - 2, 1, 1), np.nan)


def same(a, b):
    if len(a) != len(b) or isinstance(a[-1], str) or isinstance(b[-1], str):
        return False
    if a[0].shape != b[0].shape or not np.array_equal(a[0], b[0]):
        return False
    return True


def run_oracle(O, vol, units):
    try:
        xyzos, sd, R = O.detect(vol, units)
    except orc.ReferenceFails:
        return ("detect failed",)
    bins = None
    if len(xyzos):
        bins, _ = O.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
    return xyzos, sd, R, bins


def main():
    ref = orc.load_ref()
    emu = emu_lib()
    O = orc.Oracle()
    for dims, units in (((32, 32, 32), (1, 1, 1)), ((40, 36, 28), (1, 0.8, 2)), ((72, 68, 66), (1, 1, 1))):
        for name, vol in cases(dims, 1):
            r = run(ref, vol, units)
            e = run(emu, vol, units)
            q = run_oracle(O, vol, units)
            okq = same(r, q) or (isinstance(r[0], str) and isinstance(q[0], str))
            if okq and not isinstance(r[0], str) and len(r[0]) and isinstance(r[3], np.ndarray):
                okq = np.array_equal(np.isnan(r[3]), np.isnan(q[3])) and np.allclose(np.nan_to_num(r[3]), np.nan_to_num(q[3]), rtol=1e-4, atol=1e-7)
            if isinstance(r[0], str) and isinstance(e[0], str):
                print(f"{dims} {units} {name}: ref and emu both FAIL the call; oracle {'agrees' if okq else 'DIFFERS'}", flush=True)
                continue
            kr = len(r[0]) if not isinstance(r[0], str) else r[0]
            ke = len(e[0]) if not isinstance(e[0], str) else e[0]
            msg = f"{dims} {units} {name}: [oracle {'ok' if okq else 'DIFFERS'}] ref {kr} emu {ke}"
            if same(r, e):
                msg += " keypoints equal"
                if kr:
                    Rr, Re = r[2].reshape(kr, 9), e[2].reshape(ke, 9)
                    nanR = int(np.isnan(Rr).any(1).sum())
                    okR = np.array_equal(np.isnan(Rr), np.isnan(Re)) and np.nanmax(np.abs(Rr - Re), initial=0) <= 1e-5
                    msg += f" R {'ok' if okR else 'DIFF'} (nanR {nanR})"
                    if isinstance(r[3], np.ndarray) and isinstance(e[3], np.ndarray):
                        br, be = r[3], e[3]
                        nn = np.isnan(br).any(1)
                        ok = np.array_equal(np.isnan(br), np.isnan(be))
                        fin = ~np.isnan(br)
                        rel = np.abs(br[fin] - be[fin]) <= 1e-4 * np.maximum(np.abs(br[fin]), np.abs(be[fin])) + 1e-7
                        msg += f" desc nanrows {int(nn.sum())} nanpattern {'ok' if ok else 'DIFF'} within1e-4 {bool(rel.all())}"
                    else:
                        msg += f" desc ref={type(r[3]).__name__ if not isinstance(r[3], str) else r[3]} emu={type(e[3]).__name__ if not isinstance(e[3], str) else e[3]}"
            else:
                msg += "  <<< DIFFER"
            print(msg, flush=True)


if __name__ == "__main__":
    main()
