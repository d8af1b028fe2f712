#!/usr/bin/env python3
"""Per-kernel machine-code identity of csrc/s3d_gauss.hip across commits: compiles the file as it was at each given commit
(with that commit's headers) for gfx950 and prints, per fused-Gaussian kernel, whether its code bytes are the same as at the
first commit.  Backs statements like "the aligned instantiations are unchanged instruction for instruction".
usage: scripts/isa_diff.py <commit> <commit> ...   (HEAD = the working tree)"""
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift3d_amd import codeobj          # noqa: E402
from sift3d_amd import build as B       # noqa: E402


def build_at(commit, tmp):
    d = os.path.join(tmp, commit.replace("/", "_"))
    os.makedirs(d)
    if commit == "WORKTREE":
        src = os.path.join(ROOT, "sift3d_amd", "csrc")
        inc = os.path.join(ROOT, "include")
    else:
        subprocess.run(f"git -C {ROOT} archive {commit} sift3d_amd/csrc include | tar -x -C {d}", shell=True, check=True)
        src = os.path.join(d, "sift3d_amd", "csrc")
        inc = os.path.join(d, "include")
    o = os.path.join(d, "s3d_gauss.o")
    flags = [f for f in B.HIP_FLAGS if not f.startswith("-I")] + [f"-I{inc}", f"-I{src}"]
    subprocess.run([B.HIPCC, *flags, "-c", os.path.join(src, "s3d_gauss.hip"), "-o", o], check=True)
    return {k: hashlib.sha256(v).hexdigest()[:16] + f" ({len(v)} B)" for k, v in codeobj.kernel_isa(o, codeobj.GAUSS_KERNELS).items()}, \
        codeobj.kernel_isa_sha256(o, codeobj.GAUSS_KERNELS)


def main():
    commits = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        res = [build_at(c, tmp) for c in commits]
    print("| kernel | " + " | ".join(commits) + " |\n|---|" + "---|" * len(commits))
    names = sorted(set().union(*[r[0].keys() for r in res]))
    for n in names:
        base = res[0][0].get(n)
        cells = [r[0].get(n, "absent") for r in res]
        print(f"| `{n}` | " + " | ".join(c if i == 0 else ("same" if c == base else c) for i, c in enumerate(cells)) + " |")
    print("\ndigest over all of them: " + " | ".join(r[1][:16] for r in res))


if __name__ == "__main__":
    main()
