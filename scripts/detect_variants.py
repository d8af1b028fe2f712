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
variants = [("tile3 off, tables off", 0, 0), ("tile3 64^3, tables off", -1, 0), ("tile3 64^3, tables one kernel", -1, 1),
            ("tile3 64^3, tables split (default)", -1, 2)]
if os.environ.get("MORE"):
    variants += [("tile3 128^3, tables split", 128 ** 3, 2), ("tile3 32^3, tables split", 32 ** 3, 2)]
ref = None
best = {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for name, t3, om in variants:
        D.s3d_k_gauss_set_tile3(t3); D.s3d_k_set_orient_mode(om)
        s = abi.SIFT3D(); assert L.init_SIFT3D(C.byref(s)) == 0
        kp = abi.Keypoint_store(); L.init_Keypoint_store(C.byref(kp))
        ts = []
        for i in range(10):
            dev.sync(); t0 = time.perf_counter()
            assert L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp)) == 0
            dev.sync(); ts.append(time.perf_counter() - t0)
        K = kp.slab.num
        dig = "-"
        if rnd == 0:
            h = hashlib.sha256()
            for i in range(K):
                k = kp.buf[i]
                h.update(np.array([k.xd, k.yd, k.zd, k.o, k.s], np.float64).tobytes()); h.update(bytes(k.r_data))
            dig = h.hexdigest()[:16]
            if ref is None: ref = dig
            dig += " same" if dig == ref else " DIFFERENT"
        best[name] = min(best.get(name, 1e9), min(ts[2:]))
        print("round %d %-40s detect min %.3f ms median %.3f ms K=%d keypoints %s" % (rnd, name, min(ts[2:]) * 1e3, sorted(ts[2:])[4] * 1e3, K, dig), flush=True)
        L.cleanup_Keypoint_store(C.byref(kp)); L.cleanup_SIFT3D(C.byref(s))
for name, _, _ in variants:
    print("best %-40s %.3f ms" % (name, best[name] * 1e3))
