#!/usr/bin/env python3
"""A/B builds of the library that differ only in one csrc/<file>.hip, taken from git revisions (or the working tree:
rev = WORK), optionally with extra compiler flags after a colon:
    python scripts/build_file_variants.py <file> name=rev[:-DX=1,-DY=2] ...   ->  sift3d_amd/lib/ablate/libsift3d_amd_g<name>.so
(to be timed against each other on the GPU box: SIFT3D_AMD_LIB=... python scripts/gauss_time.py / match_ab.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift3d_amd import build as b   # noqa: E402

b.build()
out_dir = os.path.join(b.LIB, "ablate")
os.makedirs(out_dir, exist_ok=True)
objs = [os.path.join(b.OBJ, f) for f in os.listdir(b.OBJ) if f.endswith(".o") and not f.endswith(".testing.o") and f != sys.argv[1] + ".o"]
FILE = sys.argv[1]
for spec in sys.argv[2:]:
    name, rev = spec.split("=", 1)
    rev, _, extra = rev.partition(":")
    extra = [x for x in extra.split(",") if x]
    src = os.path.join(out_dir, f"{FILE}_{name}.hip")
    with open(src, "wb") as f:
        if rev == "WORK":
            f.write(open(os.path.join(ROOT, "sift3d_amd", "csrc", f"{FILE}.hip"), "rb").read())
        else:
            f.write(subprocess.run(["git", "show", f"{rev}:sift3d_amd/csrc/{FILE}.hip"], cwd=ROOT, check=True, capture_output=True).stdout)
    o = os.path.join(out_dir, f"{FILE}_{name}.o")
    subprocess.run([b.HIPCC, *b.HIP_FLAGS, *b.EXTRA_HIP_FLAGS.get(FILE + ".hip", []), *extra, "-x", "hip", "-c", src, "-o", o], check=True, capture_output=True)
    so = os.path.join(out_dir, f"libsift3d_amd_g{name}.so")
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so, *objs, o, "-lm", "-lz", "-lpthread", "-ldl"],
                   check=True, capture_output=True)
    print(so)
