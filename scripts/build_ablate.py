#!/usr/bin/env python3
"""Profiling builds of the library with -DDW_ABLATE=n in s3d_keypoint.hip (pieces of the descriptor kernel switched off:
1 rows + scan only, 2 no LDS atomics, 3 no back end) -> sift3d_amd/lib/ablate/libsift3d_amd_a<n>.so.  Results of these
builds are WRONG by construction; they exist to be timed (SIFT3D_AMD_LIB=... python scripts/describe_ab.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift3d_amd import build as b   # noqa: E402

b.build()
out_dir = os.path.join(b.LIB, "ablate")
os.makedirs(out_dir, exist_ok=True)
objs = [os.path.join(b.OBJ, f) for f in os.listdir(b.OBJ) if f.endswith(".o") and f != "s3d_keypoint.o"]
for n in sys.argv[1:] or ["1", "2", "3"]:
    # "3" -> -DDW_ABLATE=3;  "w512:-DDW_THREADS=512,-DDW_NCOPY=8,..." -> a variant named w512 with those defines
    if ":" in n:
        n, defs = n.split(":", 1)
        define = defs.split(",")
    else:
        define = [f"-DDW_ABLATE={n}"]
    o = os.path.join(out_dir, f"s3d_keypoint_a{n}.o")
    subprocess.run([b.HIPCC, *b.HIP_FLAGS, *b.EXTRA_HIP_FLAGS.get("s3d_keypoint.hip", []), *define, "-c",
                    os.path.join(b.CSRC, "s3d_keypoint.hip"), "-o", o], check=True, capture_output=True)
    so = os.path.join(out_dir, f"libsift3d_amd_a{n}.so")
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so, *objs, o, "-lm", "-lz",
                    "-lpthread", "-ldl"], check=True, capture_output=True)
    print(so)
