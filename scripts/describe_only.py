#!/usr/bin/env python3
"""GPU-box helper: one detect, then k_describe three times on the 512^3 bench volume (for rocprofv3 --pmc runs)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                  # noqa: E402
from sift3d_amd import abi, synth                  # noqa: E402

lib = sift3d_amd.load()
dev = sift3d_amd.load_device()
n = int(os.environ.get("N", "512"))
d_vol = dev.upload(synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0))
s = abi.SIFT3D()
assert lib.sift.init_SIFT3D(C.byref(s)) == 0
kp = abi.Keypoint_store()
lib.sift.init_Keypoint_store(C.byref(kp))
d = C.c_void_p()
assert lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp)) == 0
for _ in range(3):
    assert lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d)) == 0
dev.sync()
print("keypoints", kp.slab.num)
