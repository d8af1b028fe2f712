#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[4] (two 512^3 volumes with anisotropic units (1, 1, 1.5): detect + describe
both, then SIFT3D_nn_match) from the UNMODIFIED reference (oracle/_ref).  The inputs are the ones bench.py's
`two_volume_match` extra uses: A = the rank-0 bench volume, B = A rolled by (1, -2, 3) voxels along (z, y, x).
About 15 minutes on 8 cores; build container only:

    python tests/golden/make_golden_pair512.py          # writes tests/golden/pair512.npz
    python tests/golden/make_golden_pair512.py affine   # writes tests/golden/pair512_affine.npz: SURVEY 8d's form of the
                                                        # config -- units (1, 1, 2) (half-voxel taps along z at octave 0)
                                                        # and B = the scene of A through a known affine map (blob centres
                                                        # transformed in the generator, no resampling)

Contents (data only): per volume xyzos int16 [K,5], sd, R float32 [K,9], proj float64 [K,2] (the +-1 projections of
make_golden_512.py); match int32 [K_a] (index into B or -1, nn_thresh 0.8); sha256 of A.
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

N = int(os.environ.get("S3D_GOLDEN_N", "512"))
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "roll"
UNITS = (1.0, 1.0, 1.5) if VARIANT == "roll" else (1.0, 1.0, 2.0)
ROLL = (1, -2, 3)


def affine_tform(n):
    """Rotation by 3 degrees about z and 2 degrees about x around the volume centre, then a shift of (3.5, -2.25, 1.75)
    voxels: 3 x 4, voxel coordinates (x, y, z)."""
    a, b = np.deg2rad(3.0), np.deg2rad(2.0)
    rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    r = rx @ rz
    c = np.full(3, (n - 1) / 2.0)
    t = c - r @ c + np.array([3.5, -2.25, 1.75])
    return np.concatenate([r, t[:, None]], axis=1)


def main():
    ref = orc.load_ref()
    signs = np.random.default_rng(20260927).integers(0, 2, size=(768, 2)).astype(np.float64) * 2.0 - 1.0
    a = synth.blobs(N, N, N, synth.default_nblobs(N, N, N), seed=0)
    out, stores = {}, []
    tform = affine_tform(N)
    b = np.roll(a, ROLL, axis=(0, 1, 2)).copy() if VARIANT == "roll" else \
        synth.blobs(N, N, N, synth.default_nblobs(N, N, N), seed=0, tform=tform)
    for tag, vol in (("a", a), ("b", b)):
        t0 = time.time()
        s, im, kp = parity.run_detect(ref, vol, UNITS)
        xyzos, sd, R = ref.keypoints_to_numpy(kp)
        d = abi.SIFT3D_Descriptor_store()
        ref.sift.init_SIFT3D_Descriptor_store(C.byref(d))
        assert ref.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        bins, _ = ref.descriptors_to_numpy(d)
        print(f"volume {tag}: {len(xyzos)} keypoints, {time.time() - t0:.0f} s", flush=True)
        assert np.abs(xyzos).max() < 32768
        out.update({f"xyzos_{tag}": xyzos.astype(np.int16), f"sd_{tag}": sd.astype(np.float64),
                    f"R_{tag}": R.reshape(len(R), 9).astype(np.float32), f"proj_{tag}": bins.astype(np.float64) @ signs})
        stores.append(d)
        ref.sift.cleanup_SIFT3D(C.byref(s))
    t0 = time.time()
    ref.sift.SIFT3D_nn_match.argtypes = [C.POINTER(abi.SIFT3D_Descriptor_store), C.POINTER(abi.SIFT3D_Descriptor_store),
                                         C.c_float, C.POINTER(C.POINTER(C.c_int))]
    m = C.POINTER(C.c_int)()
    assert ref.sift.SIFT3D_nn_match(C.byref(stores[0]), C.byref(stores[1]), 0.8, C.byref(m)) == 0
    match = np.array([m[i] for i in range(stores[0].num)], np.int32)
    print(f"match: {(match >= 0).sum()} matches, {time.time() - t0:.0f} s", flush=True)
    name = ("pair%d" % N) + ("" if VARIANT == "roll" else "_" + VARIANT) + ".npz"
    path = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(path, match=match, n=np.int64(N), units=np.array(UNITS), roll=np.array(ROLL), tform=tform,
                        variant=np.array(VARIANT),
                        sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
