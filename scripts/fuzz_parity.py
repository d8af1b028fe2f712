#!/usr/bin/env python3
"""GPU-box helper: random shapes / voxel sizes / blob densities through tests.parity.check_detect_describe (keypoints,
every pyramid level and candidate counts bit-exact, descriptors within 1e-4) and check_dense, against the CPU oracle.
usage: python scripts/fuzz_parity.py [seconds] [seed]"""
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
lib = sift3d_amd.load()
O = orc.Oracle()
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
