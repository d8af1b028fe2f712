#!/usr/bin/env python3
"""GPU-box helper: one axis pass of the separable filter, REPS launches, HIP events.  env: AXES (e.g. "012"), UF, WIDTHS,
DIMS (nx,ny,nz), MODE (s3d_k_gauss_set_mode), SIFT3D_AMD_LIB (library variant)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                     # noqa: E402
from sift3d_amd import abi            # noqa: E402

REPS = int(os.environ.get("REPS", "20"))
dims = tuple(int(v) for v in os.environ.get("DIMS", "512,512,512").split(","))
uf = np.float32(eval(os.environ.get("UF", "1/0.7")))
axes = [int(c) for c in os.environ.get("AXES", "012")]
sig = {5: 0.538701, 7: 0.973294, 9: 1.22627, 11: 1.54501, 13: 1.94659, 17: 2.45255, 19: 2.8284}
widths = [int(w) for w in os.environ.get("WIDTHS", "5,9,13,17").split(",")]
dev = sift3d_amd.load_device()
lib = sift3d_amd.load()
L = dev.L
L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
L.s3d_k_gauss_set_mode(int(os.environ.get("MODE", "0")))
nx, ny, nz = dims
vol = np.random.default_rng(0).standard_normal((nz, ny, nx)).astype(np.float32)
d_src, d_dst = dev.upload(vol), dev.malloc(vol.nbytes)
out = []
for w in widths:
    g = abi.Gauss_filter()
    assert lib.imutil.init_Gauss_filter(C.byref(g), sig[w], 3) == 0 and g.f.width == w
    taps = np.ctypeslib.as_array(g.f.kernel, shape=(w,)).copy()
    for a in axes:
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.s3d_rt_event_create(C.byref(e0)); L.s3d_rt_event_create(C.byref(e1))
        dev.conv_axis(d_src, d_dst, nx, ny, nz, 1, a, taps, uf)
        dev.sync()
        L.s3d_rt_event_record(e0, None)
        for _ in range(REPS):
            dev.conv_axis(d_src, d_dst, nx, ny, nz, 1, a, taps, uf)
        L.s3d_rt_event_record(e1, None)
        ms = C.c_float()
        L.s3d_rt_event_elapsed_ms(e0, e1, C.byref(ms))
        out.append(f"w{w}{'xyz'[a]} {ms.value / REPS:.3f}")
print(os.path.basename(os.environ.get("SIFT3D_AMD_LIB", "default")), f"uf {uf:.4f}", "  ".join(out), flush=True)
