#!/usr/bin/env python3
"""GPU-box helper: N detects on one synthetic volume (env DIMS, UNITS, MODE, REPS) -- for rocprofv3 kernel traces."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                     # noqa: E402
from sift3d_amd import abi, synth     # noqa: E402

dims = tuple(int(v) for v in os.environ.get("DIMS", "512,512,512").split(","))
units = tuple(float(v) for v in os.environ.get("UNITS", "1,1,1").split(","))
dev = sift3d_amd.load_device()
lib = sift3d_amd.load()
dev.L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
dev.L.s3d_k_gauss_set_mode(int(os.environ.get("MODE", "0")))
nx, ny, nz = dims
vol = synth.blobs(nx, ny, nz, synth.default_nblobs(nx, ny, nz), 0)
d_vol = dev.upload(vol)
s = abi.SIFT3D(); lib.sift.init_SIFT3D(C.byref(s))
kp = abi.Keypoint_store(); lib.sift.init_Keypoint_store(C.byref(kp))
ts = []
for _ in range(int(os.environ.get("REPS", "3"))):
    dev.sync(); t0 = time.perf_counter()
    assert lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), nx, ny, nz, *units, C.byref(kp)) == 0
    dev.sync(); ts.append(time.perf_counter() - t0)
print(f"detect dims {dims} units {units}: {min(ts[1:]) * 1e3:.2f} ms  K {kp.slab.num}", flush=True)
