#!/usr/bin/env python3
"""Variant builds of the library that differ in the compile-time switches of s3d_keypoint.hip
("name:-DDW_THREADS=512,-DDW_NCOPY=8,..." or "nofma:-DDW_NO_FMA") -> sift3d_amd/lib/ablate/libsift3d_amd_a<name>.so, to be
timed against each other on the GPU box (scripts/describe_variants.sh; SIFT3D_AMD_LIB=... python scripts/describe_ab.py).
The ablation blocks of round 2 (-DDW_ABLATE=n: pieces of the kernel switched off, profiles/r02_describe_ablations.txt)
no longer exist in the kernel source."""
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
for n in sys.argv[1:]:
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
