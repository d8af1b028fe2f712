#!/usr/bin/env python3
"""GPU-box helper: time of one axis pass (s3d_k_conv_axis) by the table-driven kernels (mode 0) against the kernels they
replace (mode 16), per axis / tap spacing / width, HIP events around REPS back-to-back launches; and whole detects on
anisotropic and ragged volumes.  Output is a table for profiles/."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                     # noqa: E402
from sift3d_amd import abi, synth     # noqa: E402
from oracle import oracle as orc      # noqa: E402  (tap values only; nothing timed)

REPS = int(os.environ.get("REPS", "10"))
dev = sift3d_amd.load_device()
lib = sift3d_amd.load()
L = dev.L
L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
vp = C.c_void_p


def ev():
    e = vp()
    assert L.s3d_rt_event_create(C.byref(e)) == 0
    return e


def time_pass(d_src, d_dst, dims, axis, taps, uf):
    e0, e1 = ev(), ev()
    nx, ny, nz = dims
    dev.conv_axis(d_src, d_dst, nx, ny, nz, 1, axis, taps, uf)       # warm (builds the table)
    dev.sync()
    L.s3d_rt_event_record(e0, None)
    for _ in range(REPS):
        dev.conv_axis(d_src, d_dst, nx, ny, nz, 1, axis, taps, uf)
    L.s3d_rt_event_record(e1, None)
    ms = C.c_float()
    L.s3d_rt_event_elapsed_ms(e0, e1, C.byref(ms))
    L.s3d_rt_event_destroy(e0); L.s3d_rt_event_destroy(e1)
    return ms.value / REPS


def passes():
    O = orc.Oracle()
    sig = {5: 0.538701, 9: 1.22627, 13: 1.94659, 17: 2.45255}
    for dims in ((512, 512, 512), (511, 509, 303)):
        nx, ny, nz = dims
        n = nx * ny * nz
        vol = np.random.default_rng(0).standard_normal((nz, ny, nx)).astype(np.float32)
        d_src = dev.upload(vol)
        d_dst = dev.malloc(vol.nbytes)
        print(f"# dims {dims}: ms per axis pass, table-driven (mode 0) | replaced kernels (mode 16) | GB/s of 8 B/voxel (mode 0)")
        for uf in (1.0, 1.0 / 0.7, 1.0 / 1.5, 0.5, 1.25):
            if uf == 1.0 and nx % 4 == 0:
                continue
            for width, s in sig.items():
                taps = np.ascontiguousarray(O.gauss_taps(s), np.float32)
                assert taps.size == width
                row = []
                for axis in (0, 1, 2):
                    t = []
                    for mode in (0, 16):
                        L.s3d_k_gauss_set_mode(mode)
                        t.append(time_pass(d_src, d_dst, dims, axis, taps, np.float32(uf)))
                    row.append(t)
                L.s3d_k_gauss_set_mode(0)
                print(f"uf {uf:.4f} width {width:2d}: " + "  ".join(
                    f"{'xyz'[a]} {row[a][0]:.3f} | {row[a][1]:.3f} | {8e-6 * n / row[a][0]:.0f}" for a in range(3)), flush=True)
        dev.free(d_src); dev.free(d_dst)


def detects(modes=(0, 16)):
    cases = [((512, 512, 512), (1.0, 1.0, 1.0)), ((512, 512, 300), (0.7, 0.7, 1.5)), ((512, 512, 512), (1.0, 0.8, 2.0)),
             ((512, 512, 512), (1.0, 1.0, 1.5)), ((511, 509, 303), (1.0, 1.0, 1.0)), ((510, 510, 510), (1.0, 1.0, 1.0))]
    for dims, units in cases:
        nx, ny, nz = dims
        vol = synth.blobs(nx, ny, nz, synth.default_nblobs(nx, ny, nz), 0)
        d_vol = dev.upload(vol)
        for mode in modes:
            L.s3d_k_gauss_set_mode(mode)
            s = abi.SIFT3D(); lib.sift.init_SIFT3D(C.byref(s))
            kp = abi.Keypoint_store(); lib.sift.init_Keypoint_store(C.byref(kp))
            ts = []
            for _ in range(4):
                dev.sync(); t0 = time.perf_counter()
                rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), nx, ny, nz, *units, C.byref(kp))
                dev.sync(); ts.append(time.perf_counter() - t0)
                assert rc == 0
            ms = min(ts[1:]) * 1e3
            print(f"detect dims {dims} units {units} mode {mode:2d}: {ms:.2f} ms = {ms / (nx * ny * nz) * 1e6:.1f} ns/kvox  K {kp.slab.num}",
                  flush=True)
            lib.sift.cleanup_SIFT3D(C.byref(s))
        L.s3d_k_gauss_set_mode(0)
        dev.free(d_vol)


if __name__ == "__main__":
    what = sys.argv[1:] or ["passes", "detects"]
    if "passes" in what:
        passes()
    if "detects" in what:
        detects(tuple(int(m) for m in os.environ.get("MODES", "0,16").split(",")))
