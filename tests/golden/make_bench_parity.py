#!/usr/bin/env python3
"""GPU box: writes tests/golden/bench_parity.json -- the single-GPU keypoint lists (count + SHA-256, bench.kp_digest) of the
volumes `bench.py --gpus N` times, which every N > 1 line is checked against (config.parity).  The lists are THIS library's
single-GPU results; tests/test_gpu_parity.py pins the single-GPU path to the unmodified reference at 512^3 and below, and
profiles/r06_ref_1024.json (where present) pins the 1024^3 count and hash to the reference itself.
usage: python tests/golden/make_bench_parity.py [out.json]   (needs ~40 GB of HBM for the 1024^3 pyramid)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sift3d_amd                                   # noqa: E402
from sift3d_amd import synth                        # noqa: E402
from sift3d_amd import slab as S                    # noqa: E402
import bench                                        # noqa: E402

sift3d_amd.load()                                  # (binds the argument types of the plain entry points)
L = S.bind(sift3d_amd.cdll())
dev = sift3d_amd.load_device()
out = {"generator": "tests/golden/make_bench_parity.py", "volumes": {}}
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = ""
for dims in ((512, 512, 512), (1024, 1024, 1024), (512, 512, 1024), (512, 512, 2048), (512, 512, 4096)):
    k, sha = bench.single_gpu_digest(L, dev, dims, synth.default_nblobs(*dims))
    out["volumes"]["x".join(str(d) for d in dims)] = {
        "keypoints": k, "kp_sha256": sha,
        "source": f"single-GPU detect of this library on an MI355X (synth.blobs seed 0, default blob count, default parameters){', commit ' + commit if commit else ''}"}
    print(dims, k, sha, flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "bench_parity.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
