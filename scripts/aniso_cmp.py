import ctypes as C, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sift3d_amd
from sift3d_amd import abi, synth
lib = sift3d_amd.load(); dev = sift3d_amd.load_device()
n = 512
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
d_vol = dev.upload(vol)
dev.L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
for units in ((1.0, 1.0, 1.5), (1.0, 0.8, 2.0)):
    for mode in (0, 2):
        dev.L.s3d_k_gauss_set_mode(mode)
        s = abi.SIFT3D(); lib.sift.init_SIFT3D(C.byref(s))
        kp = abi.Keypoint_store(); lib.sift.init_Keypoint_store(C.byref(kp))
        ts = []
        for _ in range(3):
            dev.sync(); t0 = time.perf_counter()
            lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, *units, C.byref(kp))
            dev.sync(); ts.append(time.perf_counter() - t0)
        print("units", units, "mode", mode, "detect %.2f ms" % (min(ts[1:]) * 1e3), "K", kp.slab.num, flush=True)
        lib.sift.cleanup_SIFT3D(C.byref(s))
dev.L.s3d_k_gauss_set_mode(0)
