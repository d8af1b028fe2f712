#!/usr/bin/env python3
"""GPU-box helper: bench.py's two-volume section on its own (device-resident descriptor buffers of two SIFT3D structs)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd
from sift3d_amd import abi, synth
lib = sift3d_amd.load(); dev = sift3d_amd.load_device(); L = lib.sift
n = 512
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
d_vol = dev.upload(vol); d_vol2 = dev.upload(np.roll(vol, (1, -2, 3), axis=(0, 1, 2)))
s3, s4 = abi.SIFT3D(), abi.SIFT3D()
assert L.init_SIFT3D(C.byref(s3)) == 0 and L.init_SIFT3D(C.byref(s4)) == 0
kp3, kp4 = abi.Keypoint_store(), abi.Keypoint_store()
L.init_Keypoint_store(C.byref(kp3)); L.init_Keypoint_store(C.byref(kp4))
d3, d4 = C.c_void_p(), C.c_void_p()
for rep in range(3):
    L.sift3d_amd_detect_keypoints_dev(C.byref(s3), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.5, C.byref(kp3))
    L.sift3d_amd_extract_descriptors_dev(C.byref(s3), C.byref(kp3), C.byref(d3))
    L.sift3d_amd_detect_keypoints_dev(C.byref(s4), C.c_void_p(d_vol2), n, n, n, 1.0, 1.0, 1.5, C.byref(kp4))
    L.sift3d_amd_extract_descriptors_dev(C.byref(s4), C.byref(kp4), C.byref(d4))
    dev.sync(); t1 = time.perf_counter()
    mm = dev.nn_match(d3.value, int(kp3.slab.num), d4.value, int(kp4.slab.num), 0.8, stride=776)
    print("rep", rep, "match %.1f ms" % ((time.perf_counter() - t1) * 1e3), int((mm >= 0).sum()), "matches", flush=True)
