#!/usr/bin/env python3
"""GPU-side bisection of descriptor mismatches vs the oracle (64^3, 250 blobs, seed 0)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd
from sift3d_amd import abi, synth
from oracle import oracle as orc
from tests import parity
from tests.util import rel_close
lib = sift3d_amd.load(); O = orc.Oracle()
L = lib.sift; L.s3d_k_set_variant.argtypes = [C.c_int]
vol = synth.blobs(64, 64, 64, 250, 0)
s, im, kp = parity.run_detect(lib, vol, (1, 1, 1))
xyzos, sd, R = lib.keypoints_to_numpy(kp)
O.detect(vol)
wb, wx = O.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
for name, v in (("default", 0), ("seq_face", 16), ("f64_exp", 32), ("no_queue", 64), ("seq_face+f64exp", 48), ("all", 112)):
    L.s3d_k_set_variant(v)
    for rep in range(2):
        d = abi.SIFT3D_Descriptor_store(); L.init_SIFT3D_Descriptor_store(C.byref(d))
        assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        bins, _ = lib.descriptors_to_numpy(d)
        bad = ~rel_close(bins, wb, 1e-4, 1e-7)
        print(f"{name:18s} rep {rep}: bad {bad.sum():4d}  max abs {np.abs(bins - wb).max():.3e}  max rel(top) "
              f"{(np.abs(bins - wb) / np.maximum(np.abs(wb), 1e-3)).max():.3e}", flush=True)
        if bad.any() and rep == 0:
            ks = np.unique(np.nonzero(bad)[0])
            print("   keypoints with bad bins:", ks, "per-kp max abs", [float(np.abs(bins[k]-wb[k]).max()) for k in ks])
            for k in ks[:3]:
                idx = np.nonzero(bad[k])[0]
                print("   kp", k, xyzos[k], "bins", idx[:12], "got", bins[k, idx[:6]], "want", wb[k, idx[:6]])
L.s3d_k_set_variant(0)
