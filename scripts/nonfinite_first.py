#!/usr/bin/env python3
"""GPU-box helper: what the FIRST non-finite volume of a series costs -- finite and non-finite 512^3 volumes alternate on one SIFT3D
struct, so every non-finite one follows a finite one (no early look at the input's maximum: a full first pass is wasted)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                     # noqa: E402
from sift3d_amd import abi, synth     # noqa: E402

n = int(os.environ.get("N", "512"))
dev = sift3d_amd.load_device()
lib = sift3d_amd.load()
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
bad = vol.copy()
bad[n - 1, n - 1, n - 1] = np.nan
d_fin, d_bad = dev.upload(vol), dev.upload(bad)
s = abi.SIFT3D(); lib.sift.init_SIFT3D(C.byref(s))
kp = abi.Keypoint_store(); lib.sift.init_Keypoint_store(C.byref(kp))
t = {"finite": [], "non-finite after a finite one": []}
for rnd in range(5):
    for name, d in (("finite", d_fin), ("non-finite after a finite one", d_bad)):
        dev.sync(); t0 = time.perf_counter()
        rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
        dev.sync(); t[name].append(time.perf_counter() - t0)
for name, ts in t.items():
    print(f"{n}^3 {name}: detect {min(ts[1:]) * 1e3:.2f} ms (median {sorted(ts[1:])[len(ts[1:]) // 2] * 1e3:.2f})", flush=True)
