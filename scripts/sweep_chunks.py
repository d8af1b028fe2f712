#!/usr/bin/env python3
"""GPU-box helper: time the fused Gaussian kernels at 512^3 for several marching-chunk lengths (rows per wave in
k_gauss_xy, planes per wave in k_gauss_z)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd
import bench
dev = sift3d_amd.load_device()
sig = [0.538701, 0.973294, 1.22627, 1.54501, 1.94659, 2.45255]
for cxy, cz in ((176, 176), (128, 128), (104, 104), (88, 88), (64, 64), (256, 256), (512, 512)):
    dev.L.s3d_k_gauss_set_chunks(cxy, cz)
    apps = bench.gauss_roofline(dev, 512, sig, reps=4)
    print("chunks", cxy, cz, "xy", [a["xy_ms"] for a in apps], "z", [a["z_ms"] for a in apps], flush=True)
