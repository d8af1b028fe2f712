#!/usr/bin/env python3
"""GPU-box helper: the Z-slab driver (sift3d_amd/slab.py) with ONE rank on a 512^3 volume -- its host-side
orchestration cost against the C-API path that bench.py times at N=1."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift3d_amd                                  # noqa: E402
from sift3d_amd import synth                       # noqa: E402
from sift3d_amd.slab import Comm, SlabSift3D       # noqa: E402

n = int(os.environ.get("N", "512"))
sl = SlabSift3D(sift3d_amd.cdll(), "cuda:0", Comm(None), n, n, n)
vol = torch.from_numpy(synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)).to("cuda:0")
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sl.detect(vol)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sl.describe()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("slab x1 step %d: detect %.2f ms, describe %.2f ms, K = %d" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(sl.xyzos)),
          flush=True)
