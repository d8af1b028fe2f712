#!/usr/bin/env python3
"""Golden checksum for BASELINE configs[2] (SIFT3D_extract_dense_descriptors on a 256^3 volume, dense_rotate = 0) from
the UNMODIFIED reference (oracle/_ref).  The product's dense output is bit-identical to the reference's, so the
fixture is the SHA-256 of the 256^3 x 12 float32 output (805 MB) plus a few sampled voxels for diagnosis:

    python tests/golden/make_golden_dense256.py      # writes tests/golden/dense256.json
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from tests import parity                    # noqa: E402

N = int(os.environ.get("S3D_GOLDEN_N", "256"))


def main():
    ref = orc.load_ref()
    vol = parity.dense_input((N, N, N))
    t0 = time.time()
    out = parity.run_dense(ref, vol, (1, 1, 1))
    print(f"reference dense {N}^3: {time.time() - t0:.0f} s, output {out.shape}", flush=True)
    idx = [(1, 2, 3), (N // 2, N // 3, N // 5), (N - 2, N - 3, N - 1), (N // 7, N - 5, N // 2)]
    doc = {"n": N, "input_sha256": hashlib.sha256(np.ascontiguousarray(vol).tobytes()).hexdigest(),
           "output_sha256": hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest(),
           "samples": [{"zyx": list(i), "hist_bits": [int(b) for b in out[i].view(np.uint32)]} for i in idx]}
    path = os.path.join(ROOT, "tests", "golden", f"dense{N}.json")
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
