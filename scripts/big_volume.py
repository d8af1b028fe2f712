#!/usr/bin/env python3
"""GPU-box helper: detect + describe one N^3 volume (default 1024^3, BASELINE configs[3] on a single GPU) through the
device-resident C API; prints times and the keypoint count."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                  # noqa: E402
from sift3d_amd import abi, synth                  # noqa: E402

lib = sift3d_amd.load()
dev = sift3d_amd.load_device()
n = int(os.environ.get("N", "1024"))
t0 = time.perf_counter()
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
print("synthesised %d^3 in %.1f s" % (n, time.perf_counter() - t0), flush=True)
d_vol = dev.upload(vol)
del vol
s = abi.SIFT3D()
assert lib.sift.init_SIFT3D(C.byref(s)) == 0
kp = abi.Keypoint_store()
lib.sift.init_Keypoint_store(C.byref(kp))
d_desc = C.c_void_p()
for i in range(3):
    dev.sync()
    t0 = time.perf_counter()
    rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
    dev.sync()
    t1 = time.perf_counter()
    assert rc == 0, lib.sift.sift3d_amd_last_error().decode()
    rc = lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d_desc))
    dev.sync()
    t2 = time.perf_counter()
    assert rc == 0, lib.sift.sift3d_amd_last_error().decode()
    print("%d^3 run %d: detect %.1f ms, describe %.1f ms, %d keypoints, %.0f Mvox/s" %
          (n, i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, kp.slab.num, n ** 3 / (t2 - t0) / 1e6), flush=True)
