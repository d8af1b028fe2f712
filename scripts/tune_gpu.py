#!/usr/bin/env python3
"""Profiling helper run on the GPU box: (1) sweep of the marching-chunk sizes of the fused Gaussian
kernels at 512^3, (2) ablations of k_orient / k_describe through s3d_k_set_variant.  Prints tables."""
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


def ablate():
    vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
    d_vol = dev.upload(vol)
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    d_desc = C.c_void_p()
    L = lib.sift
    L.s3d_k_set_variant.argtypes = [C.c_int]
    res = {}
    for name, v in (("normal", 0), ("orient_no_ordered_sum", 1), ("orient_fast_exp", 2), ("orient_both", 3)):
        L.s3d_k_set_variant(v)
        L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
        dev.sync()
        res["detect_" + name] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        print("detect", name, res["detect_" + name], "ms  K =", kp.slab.num, flush=True)
    L.s3d_k_set_variant(0)
    L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
    for name, v in (("normal4", 0), ("copies2", 128), ("describe_no_atomics", 4), ("describe_no_phaseB", 8)):
        L.s3d_k_set_variant(v)
        L.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d_desc))
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            L.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d_desc))
        dev.sync()
        res["describe_" + name] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        print("describe", name, res["describe_" + name], "ms", flush=True)
    L.s3d_k_set_variant(0)
    out["ablation"] = res


if "gauss" in sys.argv or len(sys.argv) == 1:
    gauss_sweep()
if "ablate" in sys.argv or len(sys.argv) == 1:
    ablate()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tune.json"), "w"))
