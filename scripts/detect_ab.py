import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sift3d_amd
from sift3d_amd import abi, synth
lib = sift3d_amd.load(); dev = sift3d_amd.load_device(); L = lib.sift
n = 512
d_vol = dev.upload(synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0))
s = abi.SIFT3D(); assert L.init_SIFT3D(C.byref(s)) == 0
kp = abi.Keypoint_store(); L.init_Keypoint_store(C.byref(kp))
ts = []
for i in range(12):
    dev.sync(); t0 = time.perf_counter()
    assert L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp)) == 0
    dev.sync(); ts.append(time.perf_counter() - t0)
print(os.environ.get("SIFT3D_AMD_LIB", "default"), "detect min %.3f ms median %.3f ms K=%d" % (min(ts[2:]) * 1e3, sorted(ts[2:])[5] * 1e3, kp.slab.num))
