#!/usr/bin/env python3
"""GPU-box helper: what a volume with ONE non-finite voxel costs -- the streaming first pass finds out, the literal second pass
answers.  512^3, device resident, NaN at the last voxel (no candidate window can hold it: the call succeeds)."""
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
dev.L.s3d_k_gauss_set_mode(int(os.environ.get("MODE", "0")))    # 16: the verbatim pass on the per-element kernel (rounds 5-)
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
for tag, edit in (("finite", None), ("NaN at the last voxel", (n - 1, n - 1, n - 1)), ("NaN at the first voxel", (0, 0, 0))):
    v = vol.copy()
    if edit:
        v[edit] = np.nan
    d_vol = dev.upload(v)
    s = abi.SIFT3D(); lib.sift.init_SIFT3D(C.byref(s))
    kp = abi.Keypoint_store(); lib.sift.init_Keypoint_store(C.byref(kp))
    ts, rc = [], 0
    for _ in range(4):
        dev.sync(); t0 = time.perf_counter()
        rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
        dev.sync(); ts.append(time.perf_counter() - t0)
    print(f"{n}^3 {tag}: detect {min(ts[1:]) * 1e3:.2f} ms, rc {rc}, K {kp.slab.num}", flush=True)
    lib.sift.cleanup_SIFT3D(C.byref(s)); dev.free(d_vol)
