#!/usr/bin/env python3
"""GPU-box helper: accuracy (vs the oracle, 128^3) and time (512^3) of k_describe under the variant bits
of s3d_k_set_variant (bit 6 = correctly rounded expf/sqrtf/division, bit 3 = no phase B, bit 2 = no LDS
atomics).  Prints one line per variant; writes gpurun_out/ablate_desc.json."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                  # noqa: E402
from sift3d_amd import abi, synth                  # noqa: E402
from oracle import oracle as orc                   # noqa: E402  (checker only)

lib = sift3d_amd.load()
dev = sift3d_amd.load_device()
L = lib.sift
L.s3d_k_set_variant.argtypes = [C.c_int]
out = {}

# ---- accuracy --------------------------------------------------------------------------------------
n = 128
vol = synth.blobs(n, n, n, 2000, 7)
O = orc.Oracle()
xyzos, sd, R = O.detect(vol)
s = abi.SIFT3D()
assert L.init_SIFT3D(C.byref(s)) == 0
im = lib.image_from_numpy(vol)
kp = abi.Keypoint_store()
L.init_Keypoint_store(C.byref(kp))
assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
gx, gsd, gR = lib.keypoints_to_numpy(kp)
assert np.array_equal(gx, xyzos)
want, _ = O.describe(gx[:, :3].astype(np.float64), gx[:, 3:5], gsd, gR)
for v in (0,):
    L.s3d_k_set_variant(v)
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    got, _ = lib.descriptors_to_numpy(d)
    err = np.abs(got.astype(np.float64) - want) / (1e-4 * np.abs(want) + 1e-7)
    rel = np.abs(got.astype(np.float64) - want)[want > 1e-3] / want[want > 1e-3]
    out[f"acc_variant_{v}"] = {"K": int(len(gx)), "max_err_over_tol": float(err.max()),
                               "max_rel_where_gt_1e-3": float(rel.max()), "mean_rel": float(rel.mean())}
    print("accuracy variant", v, out[f"acc_variant_{v}"], flush=True)
L.s3d_k_set_variant(0)

# ---- time ------------------------------------------------------------------------------------------
n = int(os.environ.get("N", "512"))
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
d_vol = dev.upload(vol)
s2 = abi.SIFT3D()
assert L.init_SIFT3D(C.byref(s2)) == 0
kp2 = abi.Keypoint_store()
L.init_Keypoint_store(C.byref(kp2))
d_desc = C.c_void_p()
L.sift3d_amd_detect_keypoints_dev(C.byref(s2), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp2))
for name, v in (("default", 0), ("no_atomics", 4), ("no_phase_B", 8), ("copies2", 128), ("copies8", 256)):
    L.s3d_k_set_variant(v)
    L.sift3d_amd_extract_descriptors_dev(C.byref(s2), C.byref(kp2), C.byref(d_desc))
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        L.sift3d_amd_extract_descriptors_dev(C.byref(s2), C.byref(kp2), C.byref(d_desc))
    dev.sync()
    out["ms_" + name] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    print("describe", name, out["ms_" + name], "ms", flush=True)
L.s3d_k_set_variant(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ablate_desc.json"), "w"), indent=1)
