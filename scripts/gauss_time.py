#!/usr/bin/env python3
"""GPU-box helper: HIP-event times of the fused Gaussian (X+Y kernel, Z kernel) per width of the default bank at 512^3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                     # noqa: E402
import bench                          # noqa: E402

dev = sift3d_amd.load_device()
if os.environ.get("GAUSS_MODE"):                      # s3d_k_gauss_set_mode bits (128: the guarded z kernel of rounds 1-5)
    dev.L.s3d_k_gauss_set_mode(int(os.environ["GAUSS_MODE"]))
apps = bench.gauss_roofline(dev, 512, [0.538701, 0.973294, 1.22627, 1.54501, 1.94659, 2.45255], reps=int(os.environ.get("REPS", "10")))
print(os.environ.get("SIFT3D_AMD_LIB", "default"))
for a in apps:
    print(f"  width {a['width']:2d}: xy {a['xy_ms']:.4f} ms  z {a['z_ms']:.4f} ms  app {a['app_GBs']:.0f} GB/s")
