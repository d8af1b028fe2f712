#!/usr/bin/env python3
"""GPU-box helper: random shapes / voxel sizes / blob densities through tests.parity.check_detect_describe (keypoints,
every pyramid level and candidate counts bit-exact, descriptors within 1e-4) and check_dense, against the CPU oracle.
usage: python scripts/fuzz_parity.py [seconds] [seed] [nonfinite]
With a third argument: NON-FINITE mode -- every volume gets NaN / +-inf voxels written into it (single voxels, runs, slabs,
bands, a masked background) and goes through SIFT3D_detect_keypoints + SIFT3D_extract_descriptors against the oracle's answer
(itself pinned to the reference on such input: tests/test_oracle_golden.py::test_nonfinite): the same failure, or the same
keypoints and descriptors (tests.parity.assert_same_nonfinite_result).  SIFT3D_AMD_FUZZ_LIB=emu runs the emulated library
(CPU) instead of the device."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                      # noqa: E402
from oracle import oracle as orc       # noqa: E402
from tests import parity               # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nonfinite = len(sys.argv) > 3
if os.environ.get("SIFT3D_AMD_FUZZ_LIB") == "emu":
    import ctypes as C
    import subprocess
    from sift3d_amd import abi
    from sift3d_amd.device import bind_extensions
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["sh", os.path.join(emu_dir, "build_emu.sh")], check=True, capture_output=True)
    _L = C.CDLL(os.path.join(emu_dir, "libsift3d_emu.so"))
    lib = abi.Sift3dLib(_L, None, "emulated")
    bind_extensions(_L)
else:
    lib = sift3d_amd.load()
O = orc.Oracle()


def spoil(vol):
    """Writes non-finite voxels into vol (in place); returns a short description."""
    nz, ny, nx = vol.shape
    what = []
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.choice(["voxel", "voxel", "run", "zslab_lo", "zslab_hi", "xband", "corner", "last", "first"])
        val = rng.choice([np.nan, np.nan, np.nan, np.inf, -np.inf])
        z, y, x = int(rng.integers(0, nz)), int(rng.integers(0, ny)), int(rng.integers(0, nx))
        if kind == "voxel":
            vol[z, y, x] = val
        elif kind == "run":
            vol[z, y, x:x + int(rng.integers(2, 9))] = val
        elif kind == "zslab_lo":
            vol[:int(rng.integers(1, max(2, nz // 4)))] = val
        elif kind == "zslab_hi":
            vol[nz - int(rng.integers(1, max(2, nz // 4))):] = val
        elif kind == "xband":
            vol[:, :, :int(rng.integers(1, max(2, nx // 5)))] = val
        elif kind == "corner":
            vol[nz - 3:, ny - 3:, nx - 3:] = val
        elif kind == "last":
            vol[-1, -1, -1] = val
        else:
            vol[0, 0, 0] = val
        what.append(f"{kind}:{val}")
    return ",".join(what)


t0 = time.time()
n = fails = 0
while time.time() - t0 < budget:
    dims = tuple(int(v) for v in rng.integers(20, 97, 3))
    if rng.random() < 0.5:
        dims = tuple((d + 3) & ~3 for d in dims)                  # the fused fast paths need nx % 4 == 0
    units = tuple(float(rng.choice([1.0, 1.0, 1.0, 0.7, 1.5, 2.0, 0.5, 1.3])) for _ in range(3))
    if rng.random() < 0.4:
        units = (1.0, 1.0, 1.0)
    nblobs = int(rng.integers(20, max(40, dims[0] * dims[1] * dims[2] // 400)))
    seed = int(rng.integers(0, 1 << 30))
    if nonfinite:
        from sift3d_amd import synth
        vol = synth.blobs(dims[0], dims[1], dims[2], nblobs, seed)
        how = spoil(vol)
        try:
            want = parity.oracle_detect_describe_or_fail(O, vol, units)
            got = parity.detect_describe_or_fail(lib, vol, units)
            k = parity.assert_same_nonfinite_result(got, want, f"{dims} {units} {how}")
            n += 1
            print("ok", dims, units, nblobs, seed, how, "-> fails as the reference does" if want is None else f"-> K = {k}", flush=True)
        except Exception as e:                                    # report and go on
            fails += 1
            print("FAIL", dims, units, nblobs, seed, how, repr(e)[:300], flush=True)
            traceback.print_exc(limit=2)
        continue
    try:
        k = parity.check_detect_describe(lib, O, dims, units, nblobs, seed)
        if min(dims) >= 24 and rng.random() < 0.3:
            parity.check_dense(lib, O, dims, units)
        n += 1
        print("ok", dims, units, nblobs, seed, "K =", k, flush=True)
    except Exception as e:                                        # report and go on
        fails += 1
        print("FAIL", dims, units, nblobs, seed, repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
print(f"{n} configurations passed, {fails} failed in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
