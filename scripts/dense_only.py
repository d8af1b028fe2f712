#!/usr/bin/env python3
"""GPU-box helper: SIFT3D_extract_dense_descriptors device to device on a 256^3 volume, 3 times
(for rocprofv3 --kernel-trace --stats)."""
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
n = int(os.environ.get("N", "256"))
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 2)
d_in = dev.upload(vol)
d_out = dev.malloc(vol.nbytes * 12)
s = abi.SIFT3D()
assert lib.sift.init_SIFT3D(C.byref(s)) == 0
ou = (C.c_double * 3)(1.0, 1.0, 1.0)
if int(os.environ.get("DENSE_CHUNKS", "0")) > 0:       # the marching passes in that many chunks (A/B runs)
    lib.sift.s3d_k_dense_set_chunks.argtypes = [C.c_int]
    lib.sift.s3d_k_dense_set_chunks(int(os.environ["DENSE_CHUNKS"]))
for i in range(4):
    dev.sync()
    t0 = time.perf_counter()
    assert lib.sift.sift3d_amd_extract_dense_dev(C.byref(s), C.c_void_p(d_in), n, n, n, 1.0, 1.0, 1.0, ou, C.c_void_p(d_out)) == 0
    dev.sync()
    print("dense %d^3: %.3f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
