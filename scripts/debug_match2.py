#!/usr/bin/env python3
"""GPU-box helper: the two-volume configuration's descriptor sets through s3d_k_nn_match2_fast directly, printing its
return code and timing (1 = declined, -1 = HIP error), and the candidate statistics of a pass-by-pass run."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd
from sift3d_amd import abi, synth
lib = sift3d_amd.load(); dev = sift3d_amd.load_device(); L = lib.sift
n = int(os.environ.get("N", "512"))
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
sets = []
for v in (vol, np.roll(vol, (1, -2, 3), axis=(0, 1, 2))):
    d_vol = dev.upload(v)
    s = abi.SIFT3D(); assert L.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store(); L.init_Keypoint_store(C.byref(kp))
    d = C.c_void_p()
    assert L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.5, C.byref(kp)) == 0
    assert L.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d)) == 0
    K = int(kp.slab.num)
    rec = dev.download(d.value, (K, 776))
    sets.append((dev.upload(rec), K, rec))
    L.cleanup_SIFT3D(C.byref(s)); dev.free(d_vol)
vp = C.c_void_p
L.s3d_k_nn_match2_fast.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
L.s3d_rt_last_error.restype = C.c_char_p
(da, na, ra), (db, nb, rb) = sets
out = [dev.malloc(8 * max(na, nb)) for _ in range(6)]
for rep in range(2):
    dev.sync(); t0 = time.perf_counter()
    rc = L.s3d_k_nn_match2_fast(da, 776, na, db, 776, nb, *out, None)
    dev.sync()
    print("match2_fast rc", rc, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "na", na, "nb", nb, (L.s3d_rt_last_error() or b"").decode(), flush=True)
print("norms a min/max", float((ra[:, :768].astype(np.float64) ** 2).sum(1).min()), float((ra[:, :768].astype(np.float64) ** 2).sum(1).max()))
print("max element", float(ra[:, :768].max()), float(rb[:, :768].max()))
