#!/usr/bin/env python3
"""Profiling helper run on the GPU box: (1) sweep of the marching-chunk sizes of the fused Gaussian
kernels at 512^3, (2) -- ablations of the descriptor kernel moved to scripts/build_ablate.py + scripts/describe_ab.py.  Prints tables."""
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
from sift3d_amd.device import _vp                  # noqa: E402

lib = sift3d_amd.load()
dev = sift3d_amd.load_device()
n = int(os.environ.get("N", "512"))
out = {}


def taps_of(sigma):
    g = abi.Gauss_filter()
    assert lib.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
    return np.ctypeslib.as_array(g.f.kernel, shape=(g.f.width,)).copy()


def gauss_sweep():
    vol = np.random.default_rng(0).standard_normal((n, n, n)).astype(np.float32)
    d_src = dev.upload(vol)
    d_dst = dev.malloc(vol.nbytes)
    d_tmp = dev.malloc(vol.nbytes)
    ev = [_vp() for _ in range(3)]
    for e in ev:
        dev.L.s3d_rt_event_create(C.byref(e))
    res = []
    dev.L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    for sigma, mode in ((0.538701, 0), (0.538701, 1), (1.22627, 0), (1.94659, 0), (1.94659, 1), (2.45255, 0), (2.45255, 1)):
        taps = taps_of(sigma)
        dev.L.s3d_k_gauss_set_mode(mode)
        print("z-kernel mode", mode)
        for cxy, cz in ((43, 43), (64, 64), (86, 86), (103, 103), (128, 128), (171, 171), (256, 256)):
            dev.L.s3d_k_gauss_set_chunks(cxy, cz)
            txy = tz = 0.0
            reps = 4
            for r in range(reps + 1):
                dev.L.s3d_k_gauss_set_events(ev[0], ev[1], ev[2])
                dev.sep_fir(d_src, d_dst, d_tmp, n, n, n, 1, (1, 1, 1), taps, path=2)
                dev.L.s3d_k_gauss_set_events(None, None, None)
                ms = C.c_float()
                dev.L.s3d_rt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                a = ms.value
                dev.L.s3d_rt_event_elapsed_ms(ev[1], ev[2], C.byref(ms))
                if r:
                    txy += a
                    tz += ms.value
            res.append((taps.size, cxy, cz, round(txy / reps, 4), round(tz / reps, 4)))
            print("gauss width %2d chunk_xy %3d chunk_z %3d : xy %.4f ms  z %.4f ms" % res[-1], flush=True)
    dev.L.s3d_k_gauss_set_chunks(176, 176)
    dev.L.s3d_k_gauss_set_mode(0)
    out["gauss_sweep"] = res
    for p in (d_src, d_dst, d_tmp):
        dev.free(p)


if "gauss" in sys.argv or len(sys.argv) == 1:
    gauss_sweep()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tune.json"), "w"))
