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


def fields(store):
    """(coords+scale+octave+level bytes [K, 40], R [K, 9] float32) of a Keypoint_store"""
    import numpy as np
    K = int(store.slab.num)
    raw = np.ctypeslib.as_array(C.cast(store.buf, C.POINTER(C.c_uint8)), shape=(K, C.sizeof(abi.Keypoint)))
    return raw[:, 72:112].copy(), raw[:, 0:36].copy().view(np.float32).reshape(K, 9)


detail = None
if rc == 0 and os.environ.get("COMPARE_GPU", "1") != "0":
    # the same volume through THIS library on the box's GPU, field by field against the reference's list
    import hashlib
    import numpy as np
    import sift3d_amd
    lib = sift3d_amd.load()
    s2 = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s2)) == 0
    kp2 = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp2))
    im2 = lib.image_from_numpy(vol)
    assert lib.sift.SIFT3D_detect_keypoints(C.byref(s2), C.byref(im2), C.byref(kp2)) == 0
    a_pos, a_R = fields(kp)
    b_pos, b_R = fields(kp2)
    detail = {"keypoints_this_library": int(kp2.slab.num)}
    if a_pos.shape == b_pos.shape:
        dR = np.abs(a_R.astype(np.float64) - b_R)
        detail.update({"position_scale_octave_level_bytes_equal": bool(np.array_equal(a_pos, b_pos)),
                       "rows_with_other_position_bytes": int((a_pos != b_pos).any(1).sum()),
                       "R_bit_equal_rows": int((a_R.view(np.uint32) == b_R.view(np.uint32)).all(1).sum()),
                       "R_max_abs_diff": float(dR.max()), "R_rows_beyond_1e-5": int((dR.max(1) > 1e-5).sum()),
                       "sha256_positions": [hashlib.sha256(a_pos.tobytes()).hexdigest(), hashlib.sha256(b_pos.tobytes()).hexdigest()]})
g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_parity.json")))["volumes"].get(f"{n}x{n}x{n}")
rec = {"n": n, "rc": rc, "keypoints": k, "kp_sha256": sha, "detect_s": round(dt, 1), "threads": os.environ.get("OMP_NUM_THREADS"),
       "library": "oracle/_ref: the unmodified reference, gcc -O3, OpenMP", "this_library": g,
       "equal": bool(g and g["keypoints"] == k and g["kp_sha256"] == sha), "field_by_field_vs_this_library_on_the_gpu": detail}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec), flush=True)
