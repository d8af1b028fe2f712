#!/usr/bin/env python3
"""GPU-box helper: time the descriptor kernel on the 512^3 bench volume (and N=VARIANTS env: comma list of
s3d_k_set_variant values to compare, e.g. "0,512"); checks that the variants agree."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                  # noqa: E402
from sift3d_amd import abi, synth                  # noqa: E402

lib = sift3d_amd.load()
dev = sift3d_amd.load_device()
L = lib.sift
n = int(os.environ.get("N", "512"))
uz = float(os.environ.get("UZ", "1.0"))
d_vol = dev.upload(synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0))
s = abi.SIFT3D()
assert L.init_SIFT3D(C.byref(s)) == 0
kp = abi.Keypoint_store()
L.init_Keypoint_store(C.byref(kp))
d = C.c_void_p()
assert L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, uz, C.byref(kp)) == 0
K = int(kp.slab.num)
has_variant = hasattr(L, "s3d_k_set_variant")
if has_variant:
    L.s3d_k_set_variant.argtypes = [C.c_int]
ref = None
for v in [int(x) for x in os.environ.get("VARIANTS", "0").split(",")]:
    if has_variant:
        L.s3d_k_set_variant(v)
    ts = []
    for _ in range(4):
        dev.sync()
        t0 = time.perf_counter()
        assert L.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d)) == 0
        dev.sync()
        ts.append(time.perf_counter() - t0)
    rec = dev.download(d.value, (K, 776))[:, :768]
    msg = ""
    if ref is None:
        ref = rec
    else:
        den = np.maximum(np.abs(ref), np.abs(rec))
        err = np.abs(ref.astype(np.float64) - rec) / (1e-4 * den + 1e-7)
        msg = f" max err/tol vs first variant {err.max():.4f}"
    print(f"variant {v}: {K} keypoints, describe {min(ts[1:]) * 1e3:.2f} ms (runs {[round(t * 1e3, 2) for t in ts]}){msg}", flush=True)
if has_variant:
    L.s3d_k_set_variant(0)
# SAVE=path stores this library's descriptors, CMP=path compares against another library's (SIFT3D_AMD_LIB=...)
if os.environ.get("SAVE"):
    np.save(os.environ["SAVE"], rec)
if os.environ.get("CMP"):
    other = np.load(os.environ["CMP"])
    den = np.maximum(np.abs(other), np.abs(rec))
    err = np.abs(other.astype(np.float64) - rec) / (1e-4 * den + 1e-7)
    print(f"against {os.environ['CMP']}: max err/tol {err.max():.4f}", flush=True)
