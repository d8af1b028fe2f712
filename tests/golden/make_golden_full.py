#!/usr/bin/env python3
"""Full-size golden vectors for the any-spacing / ragged-row paths from the UNMODIFIED reference (oracle/_ref compiled
from /root/reference where it lies), in the layout of full512.npz (see make_golden_512.py):

    python tests/golden/make_golden_full.py aniso07      # 512 x 512 x 300 voxels of 0.7 x 0.7 x 1.5
    python tests/golden/make_golden_full.py odd511       # 511 x 509 x 303 unit voxels (rows not a multiple of 4)

Several minutes of CPU each; run in the build container, never on the GPU box.  The volume is
sift3d_amd.synth.blobs(nx, ny, nz, default_nblobs, seed 0) -- what bench.py's extras and the GPU tests regenerate
(its SHA-256 is in the fixture).  Data only: keypoints, scales, orientations, level hashes, sampled descriptors,
projections of every descriptor.
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
from tests.golden.make_golden_512 import EVERY, level_hashes, projection_signs   # noqa: E402

CASES = {"aniso07": ((512, 512, 300), (0.7, 0.7, 1.5)), "odd511": ((511, 509, 303), (1.0, 1.0, 1.0))}


def main(name):
    (nx, ny, nz), units = CASES[name]
    ref = orc.load_ref()
    vol = synth.blobs(nx, ny, nz, synth.default_nblobs(nx, ny, nz), seed=0)
    t0 = time.time()
    s, im, kp = parity.run_detect(ref, vol, units)
    xyzos, sd, R = ref.keypoints_to_numpy(kp)
    print(f"{name}: reference detect: {len(xyzos)} keypoints in {time.time() - t0:.0f} s", flush=True)
    gss_sha, dog_sha = level_hashes(s.gpyr), level_hashes(s.dog)
    t0 = time.time()
    d = abi.SIFT3D_Descriptor_store()
    ref.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert ref.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, xyz = ref.descriptors_to_numpy(d)
    print(f"{name}: reference describe: {time.time() - t0:.0f} s", flush=True)
    assert np.abs(xyzos).max() < 32768
    out = os.path.join(ROOT, "tests", "golden", f"full_{name}.npz")
    np.savez_compressed(out, xyzos=xyzos.astype(np.int16), sd=sd.astype(np.float64),
                        R=R.reshape(len(R), 9).astype(np.float32),
                        proj=bins.astype(np.float64) @ projection_signs(), every=np.int64(EVERY),
                        desc=bins[::EVERY].astype(np.float32), dims=np.array([nx, ny, nz], np.int64),
                        units=np.array(units, np.float64), gss_sha=gss_sha, dog_sha=dog_sha,
                        sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(vol).tobytes()).digest(), np.uint8))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    for n in sys.argv[1:] or list(CASES):
        main(n)
