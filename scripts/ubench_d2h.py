import ctypes as C, time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sift3d_amd
dev = sift3d_amd.load_device(); L = dev.L
n = 97 * 1024 * 1024
d = dev.malloc(n)
host = np.empty(n, np.uint8); host[:] = 1
L.s3d_rt_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
L.s3d_rt_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
p = C.c_void_p(); assert L.s3d_rt_host_alloc(C.byref(p), n) == 0
for name, dst in (("pageable", host.ctypes.data), ("pinned", p.value)):
    for r in range(3):
        dev.sync(); t0 = time.perf_counter()
        assert L.s3d_rt_d2h(dst, d, n, None) == 0
        dev.sync(); t = time.perf_counter() - t0
    print(name, f"{t*1e3:.2f} ms  {n/t/1e9:.1f} GB/s")
pin = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))
t0 = time.perf_counter(); host[:] = pin; t = time.perf_counter() - t0
print("memcpy pinned->pageable", f"{t*1e3:.2f} ms {n/t/1e9:.1f} GB/s")
