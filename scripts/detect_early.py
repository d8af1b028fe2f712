"""GPU-box helper: SIFT3D detect at 512^3 with the orientation window sums of octave 0's candidates on a stream confined to
part of the CUs (SIFT3D_AMD_EARLY_ORIENT=<CUs>, SIFT3D_AMD_EARLY_ORIENT_WAVES=<waves>) beside the coarse octaves' filters;
variants interleaved over three rounds in one process, keypoints hashed (they must not move)."""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import sift3d_amd
from sift3d_amd import abi, synth
lib = sift3d_amd.load(); dev = sift3d_amd.load_device(); L = lib.sift
n = int(os.environ.get("N", "512"))
d_vol = dev.upload(synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0))
variants = [("off", 0, 0)] + [(f"{c} CUs, {w} waves", c, w) for c, w in
                              ((192, 3072), (224, 3584), (128, 2048), (192, 6144), (256, 4096), (224, 1792))]
ref, best = None, {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for name, cus, waves in variants:
        if cus: os.environ["SIFT3D_AMD_EARLY_ORIENT"] = str(cus); os.environ["SIFT3D_AMD_EARLY_ORIENT_WAVES"] = str(waves)
        else: os.environ.pop("SIFT3D_AMD_EARLY_ORIENT", None)
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
        print("round %d %-24s detect min %.3f ms median %.3f ms K=%d keypoints %s" % (rnd, name, min(ts[2:]) * 1e3, sorted(ts[2:])[4] * 1e3, K, dig), flush=True)
        L.cleanup_Keypoint_store(C.byref(kp)); L.cleanup_SIFT3D(C.byref(s))
for name, _, _ in variants:
    print("best %-24s %.3f ms" % (name, best[name] * 1e3))
