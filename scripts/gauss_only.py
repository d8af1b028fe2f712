#!/usr/bin/env python3
"""Runs each Gaussian of the default bank REPS times on a 512^3 volume in HBM (fused path) -- the
workload for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE per launch)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd
from sift3d_amd import abi
lib = sift3d_amd.load(); dev = sift3d_amd.load_device()
n = int(os.environ.get("N", "512")); reps = int(os.environ.get("REPS", "3"))
vol = np.random.default_rng(0).standard_normal((n, n, n)).astype(np.float32)
d_src = dev.upload(vol); d_dst = dev.malloc(vol.nbytes); d_tmp = dev.malloc(vol.nbytes)
for sigma in (0.538701, 0.973294, 1.22627, 1.54501, 1.94659, 2.45255):
    g = abi.Gauss_filter(); assert lib.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
    taps = np.ctypeslib.as_array(g.f.kernel, shape=(g.f.width,)).copy()
    for _ in range(reps):
        dev.sep_fir(d_src, d_dst, d_tmp, n, n, n, 1, (1, 1, 1), taps, path=2)
dev.sync()
print("done")
