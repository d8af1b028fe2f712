#!/usr/bin/env python3
"""GPU-box helper: time SIFT3D_nn_match (device-resident stores) on K x K descriptor-like rows and print a checksum of
the match indices (two builds must agree: SIFT3D_AMD_LIB=... to point at a variant)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                   # noqa: E402
from tests.util import rand_desc, match_sets        # noqa: E402

dev = sift3d_amd.load_device()
sift3d_amd.load()
K = int(os.environ.get("K", "31207"))
a = rand_desc(K, 1)
b = match_sets(a[: K // 2], 2)
b = np.vstack([b, rand_desc(K - len(b), 3)]) if len(b) < K else b[:K]
d_a, d_b = dev.upload(a), dev.upload(b)
ts = []
for _ in range(5):
    dev.sync()
    t0 = time.perf_counter()
    m = dev.nn_match(d_a, K, d_b, len(b), 0.8, stride=768)
    ts.append(time.perf_counter() - t0)
print(os.environ.get("SIFT3D_AMD_LIB", "default"), f"K={K} x {len(b)}: match min {min(ts[1:]) * 1e3:.2f} ms (runs {[round(t * 1e3, 2) for t in ts]}), "
      f"matched {int((m >= 0).sum())}, checksum {int((m.astype(np.int64) * np.arange(1, K + 1)).sum()) & 0xffffffff:08x}")
