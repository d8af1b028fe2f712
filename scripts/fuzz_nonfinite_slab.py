#!/usr/bin/env python3
"""CPU: random NaN / inf placements (single voxels, masked slabs, the seam between two ranks, first / last voxel) in volumes detected +
described on 2-3 loop-back Z-slab ranks of the EMULATED library, against the oracle (pinned to the reference on non-finite input).
usage: python scripts/fuzz_nonfinite_slab.py <seconds>"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
import numpy as np
import nan_probe as P
from oracle import oracle as orc
from sift3d_amd import synth, slab as slabmod
from tests import parity
emu = P.emu_lib(); slabmod.bind(emu.sift)
O = orc.Oracle()
rng = np.random.default_rng(11)
t0 = time.time(); n = bad = 0
while time.time() - t0 < float(sys.argv[1]):
    dims = (int(rng.integers(24, 40)), int(rng.integers(24, 40)), int(rng.choice([64, 66, 72, 96])))
    ng = 3 if dims[2] == 96 and rng.random() < 0.5 else 2
    vol = synth.blobs(dims[0], dims[1], dims[2], int(rng.integers(80, 260)), int(rng.integers(0, 1 << 30)))
    nz, ny, nx = vol.shape
    how = []
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.choice(["voxel", "voxel", "zslab_lo", "zslab_hi", "seam", "last", "first", "inf"])
        z, y, x = int(rng.integers(0, nz)), int(rng.integers(0, ny)), int(rng.integers(0, nx))
        if kind == "voxel": vol[z, y, x] = np.nan
        elif kind == "inf": vol[z, y, x] = np.inf
        elif kind == "zslab_lo": vol[:int(rng.integers(1, 6))] = np.nan
        elif kind == "zslab_hi": vol[nz - int(rng.integers(1, 6)):] = np.nan
        elif kind == "seam": vol[nz // ng - 1: nz // ng + 1, y, x] = np.nan
        elif kind == "last": vol[-1, -1, -1] = np.nan
        else: vol[0, 0, 0] = np.nan
        how.append(kind)
    want = parity.oracle_detect_describe_or_fail(O, vol, (1, 1, 1), parity.SLAB_PARAMS)
    try:
        got = parity.detect_describe_or_fail(emu, vol, (1, 1, 1), parity.SLAB_PARAMS, ngpu=ng)
        k = parity.assert_same_nonfinite_result(got, want, f"{dims} {how} on {ng} ranks")
        n += 1; print("ok", dims, ng, how, "fails" if want is None else k, flush=True)
    except AssertionError as e:
        bad += 1; print("FAIL", dims, ng, how, str(e)[:200], flush=True)
print(n, "passed", bad, "failed")
