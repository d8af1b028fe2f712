#!/usr/bin/env python3
"""GPU-box helper: the reference API with HOST buffers (SIFT3D_detect_keypoints / SIFT3D_extract_descriptors on a
512^3 Image in pageable memory): what a relinked caller sees, PCIe transfers included."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                  # noqa: E402
from sift3d_amd import abi, synth                  # noqa: E402

lib = sift3d_amd.load()
n = int(os.environ.get("N", "512"))
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
im = lib.image_from_numpy(vol)
s = abi.SIFT3D()
assert lib.sift.init_SIFT3D(C.byref(s)) == 0
kp = abi.Keypoint_store()
lib.sift.init_Keypoint_store(C.byref(kp))
d = abi.SIFT3D_Descriptor_store()
lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
for i in range(4):
    t0 = time.perf_counter()
    assert lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
    t1 = time.perf_counter()
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    t2 = time.perf_counter()
    print("host API run %d: detect %.1f ms, describe %.1f ms, total %.1f ms, K = %d" %
          (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, kp.slab.num), flush=True)
