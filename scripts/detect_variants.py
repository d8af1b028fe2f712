"""GPU-box helper: one process, the 512^3 benchmark volume, SIFT3D detect under the knobs of this round -- the one-launch
tile kernel for small octaves (s3d_k_gauss_set_tile3) and the orientation window tables (s3d_k_set_orient_mode) --
min / median wall time per variant and a hash of the keypoints (they must not move)."""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import sift3d_amd
from sift3d_amd import abi, synth
lib = sift3d_amd.load(); dev = sift3d_amd.load_device(); L = lib.sift; D = dev.L
n = int(os.environ.get("N", "512"))
d_vol = dev.upload(synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0))
D.s3d_k_gauss_set_tile3.argtypes = [C.c_long]
D.s3d_k_set_orient_mode.argtypes = [C.c_int]
variants = [("tile3 off, tables off", 0, 0), ("tile3 64^3, tables off", -1, 0), ("tile3 off, tables one kernel", 0, 1),
            ("tile3 off, tables split", 0, 2), ("tile3 64^3, tables split (default)", -1, 2),
            ("tile3 128^3, tables split", 128 ** 3, 2), ("tile3 256^3, tables split", 256 ** 3, 2),
            ("tile3 off, tables off (again)", 0, 0)]
ref = None
for name, t3, om in variants:
    D.s3d_k_gauss_set_tile3(t3); D.s3d_k_set_orient_mode(om)
    s = abi.SIFT3D(); assert L.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store(); L.init_Keypoint_store(C.byref(kp))
    ts = []
    for i in range(12):
        dev.sync(); t0 = time.perf_counter()
        assert L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp)) == 0
        dev.sync(); ts.append(time.perf_counter() - t0)
    K = kp.slab.num
    h = hashlib.sha256()
    for i in range(K):
        k = kp.buf[i]
        h.update(np.array([k.xd, k.yd, k.zd, k.o, k.s], np.float64).tobytes()); h.update(bytes(k.r_data))
    dig = h.hexdigest()[:16]
    if ref is None: ref = dig
    print("%-40s detect min %.3f ms median %.3f ms K=%d keypoints %s %s" % (name, min(ts[2:]) * 1e3, sorted(ts[2:])[5] * 1e3, K, dig,
          "same" if dig == ref else "DIFFERENT"), flush=True)
    L.cleanup_Keypoint_store(C.byref(kp)); L.cleanup_SIFT3D(C.byref(s))
