#!/usr/bin/env python3
"""Golden vectors for volumes with non-finite voxels, from the UNMODIFIED reference (oracle/_ref):

    python tests/golden/make_golden_nonfinite.py          # writes tests/golden/nonfinite.npz  (about a minute)

For every entry of tests/parity.py NONFINITE_CASES (a seeded synthetic volume with NaN / infinite voxels written into
it): whether the reference's SIFT3D_detect_keypoints fails (a NaN gradient inside a candidate's orientation window:
eigen_Mat_rm / LAPACK dsyevd, sift.c:1430), and otherwise its keypoints, scales, orientations and ALL descriptor floats.
Data only.  Keys: "<base>/<name>/fail|xyzos|sd|R|desc|sha256".
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from tests import parity                    # noqa: E402


def main():
    ref = orc.load_ref()
    out = {}
    for base, name, edits in parity.NONFINITE_CASES:
        vol, units, params = parity.nonfinite_case(base, edits)
        r = parity.detect_describe_or_fail(ref, vol, units, params)
        k = f"{base}/{name}/"
        out[k + "sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(vol).tobytes()).digest(), np.uint8)
        out[k + "fail"] = np.int64(r is None)
        if r is not None:
            out[k + "xyzos"] = r[0].astype(np.int32)
            out[k + "sd"] = r[1].astype(np.float64)
            out[k + "R"] = r[2].reshape(-1, 9).astype(np.float32)
            out[k + "desc"] = r[3].astype(np.float32)
        print(base, name, "fails" if r is None else f"{len(r[0])} keypoints", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "nonfinite.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
