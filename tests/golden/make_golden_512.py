#!/usr/bin/env python3
"""Golden vectors at BASELINE configs[1] (512^3 float32, the bench workload of rank 0) from the UNMODIFIED reference
(oracle/_ref: libimutil.so + libsift3D.so compiled from /root/reference where it lies).  Takes the reference roughly
an hour on 8 cores; run in the build container, never on the GPU box:

    python tests/golden/make_golden_512.py            # writes tests/golden/full512.npz

Contents (data only):
  xyzos   int16 [K,5]   every keypoint (x, y, z, octave, level) in the reference's order
  sd      float64 [K]   keypoint scales
  R       float32 [K,9] orientation matrices
  proj    float64 [K,16] sixteen fixed +-1 projections of every 768-float descriptor (signs from a seeded generator; the
                         first two columns are those of the first version of this fixture)
  gss_sha / dog_sha  uint8 [levels,32]  SHA-256 of every GSS / DoG level of the reference's pyramids (octave-major)
  every   int           descriptor sampling stride
  desc    float32 [ceil(K/every),768]  the descriptors of keypoints 0, every, 2*every, ...
  sha256  of the float32 input volume (so the test knows it regenerated the same input)
"""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from sift3d_amd import abi, synth           # noqa: E402
from tests import parity                    # noqa: E402

EVERY = 32
N = int(os.environ.get("S3D_GOLDEN_N", "512"))


NPROJ = 16


def projection_signs():
    first = np.random.default_rng(20260927).integers(0, 2, size=(768, 2)).astype(np.float64) * 2.0 - 1.0
    more = np.random.default_rng(20260928).integers(0, 2, size=(768, NPROJ - 2)).astype(np.float64) * 2.0 - 1.0
    return np.concatenate([first, more], axis=1)


def level_hashes(pyr):
    out = []
    for i in range(pyr.num_octaves * pyr.num_levels):
        lv = pyr.levels[i]
        a = np.ctypeslib.as_array(lv.data, shape=(lv.nx * lv.ny * lv.nz,))
        out.append(np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8))
    return np.stack(out)


def main():
    ref = orc.load_ref()
    vol = synth.blobs(N, N, N, synth.default_nblobs(N, N, N), seed=0)
    t0 = time.time()
    s, im, kp = parity.run_detect(ref, vol, (1, 1, 1))
    xyzos, sd, R = ref.keypoints_to_numpy(kp)
    print(f"reference detect: {len(xyzos)} keypoints in {time.time() - t0:.0f} s", flush=True)
    gss_sha, dog_sha = level_hashes(s.gpyr), level_hashes(s.dog)
    t0 = time.time()
    d = abi.SIFT3D_Descriptor_store()
    ref.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert ref.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, xyz = ref.descriptors_to_numpy(d)
    print(f"reference describe: {time.time() - t0:.0f} s", flush=True)
    assert np.abs(xyzos).max() < 32768
    out = os.path.join(ROOT, "tests", "golden", "full512.npz" if N == 512 else f"full{N}.npz")
    np.savez_compressed(out, xyzos=xyzos.astype(np.int16), sd=sd.astype(np.float64),
                        R=R.reshape(len(R), 9).astype(np.float32),
                        proj=bins.astype(np.float64) @ projection_signs(), every=np.int64(EVERY),
                        desc=bins[::EVERY].astype(np.float32), n=np.int64(N), gss_sha=gss_sha, dog_sha=dog_sha,
                        sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(vol).tobytes()).digest(), np.uint8))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
