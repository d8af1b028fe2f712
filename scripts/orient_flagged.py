"""GPU-box diagnostic: orientation mode 3 on a synthetic level set -- how many candidates the staged kernel hands to the
general path (the count stays at the head of the flagged list in the tables buffer), and the staged forms' headers."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import sift3d_amd
from tests import parity
lib = sift3d_amd.load(); dev = sift3d_amd.load_device(); L = dev.L
n = int(os.environ.get("N", "192"))
rng = np.random.default_rng(1)
sig = (2.0159, 2.5398, 3.2)
nl = len(sig)
vols = [rng.standard_normal((n, n, n)).astype(np.float32) for _ in range(nl)]
pd = parity._PyrDesc()
d_lv = [dev.upload(v) for v in vols]
for i, p_ in enumerate(d_lv): pd.d_level[i] = p_
pd.dims[0][0] = pd.dims[0][1] = pd.dims[0][2] = n
for a in range(3): pd.unitsf[0][a] = 1.0
pd.num_octaves, pd.num_levels, pd.first_level = 1, nl, 0
K = 100000
xs, ys, zs = (rng.integers(1, n - 1, K) for _ in range(3))
idx = (zs * n * n + ys * n + xs).astype(np.uint32)
tag = rng.integers(0, nl, K).astype(np.uint32)
d_idx, d_tag, d_sig = dev.upload(idx), dev.upload(tag), dev.upload(np.asarray(sig, np.float64))
L.s3d_k_orient_tab_bytes.restype = C.c_size_t; L.s3d_k_orient_tab_bytes.argtypes = [C.c_void_p]
L.s3d_k_orient_scratch_bytes.restype = C.c_size_t; L.s3d_k_orient_scratch_bytes.argtypes = [C.c_uint32]
tb = L.s3d_k_orient_tab_bytes(C.byref(pd))
d_tab, d_R, d_keep, d_scr = dev.malloc(tb), dev.malloc(K * 36), dev.malloc(K * 4), dev.malloc(L.s3d_k_orient_scratch_bytes(K))
L.s3d_k_orient_tab.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_double,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.s3d_k_set_orient_mode.argtypes = [C.c_int]
L.s3d_k_set_orient_mode(3)
assert L.s3d_k_orient_tab(C.byref(pd), d_idx, d_tag, None, K, d_sig, 0.4, d_R, d_keep, None, d_scr, d_tab, None) == 0
assert L.s3d_rt_sync(None) == 0
tab = parity._ORI_TAB_DT.itemsize
win = 64 + 8 * 1232 + 32 * 128 * 64
raw = dev.download(d_tab, (tb,), np.uint8)
for k in range(nl):
    h = np.frombuffer(raw[tab * nl + k * win:][:64].tobytes(), np.int32)
    print("level", k, "n_turns", h[0], "rb", h[1:7], "n_rows", h[7], "lds_floats", h[8])
cnt = np.frombuffer(raw[(tab + win) * nl:][:4].tobytes(), np.uint32)[0]
R = 10
print("flagged", cnt, "of", K, "expected about", K - int(K * ((n - 2 * 9) / (n - 2)) ** 3))
