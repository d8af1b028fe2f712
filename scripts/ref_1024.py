#!/usr/bin/env python3
"""GPU box's HOST cores: the unmodified reference (oracle/_ref) detects keypoints in BASELINE configs[3]'s volume -- 1024^3, the
bench generator, seed 0 -- and the digest of its list (bench.kp_digest) goes into a JSON next to this library's
(tests/golden/bench_parity.json).  ~60 GB of host memory, tens of minutes on 64 threads; refuses below 150 GB of free memory.
usage: OMP_NUM_THREADS=64 OPENBLAS_NUM_THREADS=1 python scripts/ref_1024.py [n=1024] [out.json]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift3d_amd import abi, synth            # noqa: E402
from oracle import oracle as orc             # noqa: E402
import bench                                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"ref_{n}.json")
free_gb = 0.0
for line in open("/proc/meminfo"):
    if line.startswith("MemAvailable:"):
        free_gb = int(line.split()[1]) / 2**20
need = 150.0 * (n / 1024.0) ** 3
print(f"MemAvailable {free_gb:.0f} GiB, cores {os.cpu_count()}, need ~{need:.0f} GiB", flush=True)
if free_gb < need:
    json.dump({"n": n, "refused": f"only {free_gb:.0f} GiB of host memory available"}, open(out, "w"))
    sys.exit(0)
assert orc.have_ref()
ref = orc.load_ref()
vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
s = abi.SIFT3D()
assert ref.sift.init_SIFT3D(C.byref(s)) == 0
im = ref.image_from_numpy(vol)
kp = abi.Keypoint_store()
ref.sift.init_Keypoint_store(C.byref(kp))
t0 = time.perf_counter()
rc = ref.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp))
dt = time.perf_counter() - t0
k, sha = bench.kp_digest(kp) if rc == 0 else (-1, "")
g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_parity.json")))["volumes"].get(f"{n}x{n}x{n}")
rec = {"n": n, "rc": rc, "keypoints": k, "kp_sha256": sha, "detect_s": round(dt, 1), "threads": os.environ.get("OMP_NUM_THREADS"),
       "library": "oracle/_ref: the unmodified reference, gcc -O3, OpenMP", "this_library": g,
       "equal": bool(g and g["keypoints"] == k and g["kp_sha256"] == sha)}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec), flush=True)
