"""Build libsift3d_amd.so (HIP kernels for gfx950 + host C) and libs3d_synth.so, in-tree.

    python -m sift3d_amd.build            # or __graft_entry__.build()

Host code is C (gcc -std=gnu11); device code is HIP (hipcc --offload-arch=gfx950).  Every translation
unit that does float arithmetic on the parity path is compiled with -ffp-contract=off: the reference
build has no fused multiply-add and contraction changes pyramid bits (hard part 1).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "lib", "obj")
INC = os.path.join(ROOT, "include")

HIP_SOURCES = ["s3d_rt.hip", "s3d_image.hip", "s3d_gauss.hip", "s3d_gauss_tab.hip", "s3d_extrema.hip", "s3d_keypoint.hip",
               "s3d_dense.hip", "s3d_match.hip", "s3d_resample.hip", "s3d_rccl.hip"]
C_SOURCES = ["host/s3d_host_util.c", "host/s3d_host_api.c", "host/s3d_host_match.c", "host/s3d_host_io.c", "host/s3d_host_cli.c", "host/s3d_host_reg.c", "host/s3d_host_draw.c", "host/s3d_host_slab.c", "host/s3d_host_mat.c"]
BIN = os.path.join(HERE, "bin")
CLI_PROGRAMS = ["kpSift3D", "denseSift3D", "regSift3D"]
# Per-file extra flags.  s3d_keypoint.hip: the SLP vectoriser pairs scalar f32 operations into v_pk_* instructions, which on
# gfx950 cost 4.4 cycles per wave64 against 2.8 for a scalar f32 op (scripts/ubench_valu.hip) and need register shuffles
# and s_nops around them: a net loss for the VALU-bound descriptor kernel.
# s3d_gauss_tab.hip: its packed operations are written out (s3d_f2); what the vectoriser adds on top pairs values of
# different taps and pays for it in register moves and s_nops.
EXTRA_HIP_FLAGS = {"s3d_keypoint.hip": ["-fno-slp-vectorize"], "s3d_gauss_tab.hip": ["-fno-slp-vectorize"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-function", f"-I{INC}", f"-I{CSRC}"]
C_FLAGS = ["-std=gnu11", "-O2", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra", f"-I{INC}", f"-I{CSRC}/host",
           "-pthread"]


def _newer(src: str, out: str, extra=()) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + cmd[-1])


# Sources with diagnostic switches / test hooks behind -DS3D_TESTING (csrc/host/s3d_host.h): compiled a second time for
# lib/libsift3d_amd_testing.so, which differs from the product library in these objects only.  The product library has
# neither sift3d_amd_slab_test_inject nor the S3D_* environment switches.
TESTING_SOURCES = ["s3d_gauss.hip", "s3d_gauss_tab.hip", "s3d_extrema.hip", "s3d_keypoint.hip", "host/s3d_host_api.c", "host/s3d_host_match.c",
                   "host/s3d_host_slab.c"]


def build(verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(INC, h) for h in os.listdir(INC)] + \
              [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + \
              [os.path.join(CSRC, "host", h) for h in os.listdir(os.path.join(CSRC, "host")) if h.endswith(".h")]
    # every stale translation unit of both libraries is compiled in one pool (a clean build is bounded by the longest
    # single file, s3d_gauss.hip with its 9 x 6 fused-kernel instantiations: ~75 s, instead of the sum: ~7 minutes)
    objs, jobs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(OBJ, s.replace(".hip", ".o"))
        if _newer(src, o, headers):
            jobs.append(("hipcc " + s, [HIPCC, *HIP_FLAGS, *EXTRA_HIP_FLAGS.get(s, []), "-c", src, "-o", o]))
        objs.append(o)
    for s in C_SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(OBJ, os.path.basename(s).replace(".c", ".o"))
        if _newer(src, o, headers):
            jobs.append(("gcc " + s, ["gcc", *C_FLAGS, "-c", src, "-o", o]))
        objs.append(o)
    tobjs = list(objs)
    for s in TESTING_SOURCES:
        src = os.path.join(CSRC, s)
        base = os.path.basename(s).rsplit(".", 1)[0]
        o = os.path.join(OBJ, base + ".testing.o")
        if _newer(src, o, headers):
            if s.endswith(".hip"):
                jobs.append(("testing build: " + s, [HIPCC, *HIP_FLAGS, *EXTRA_HIP_FLAGS.get(s, []), "-DS3D_TESTING", "-c", src, "-o", o]))
            else:
                jobs.append(("testing build: " + s, ["gcc", *C_FLAGS, "-DS3D_TESTING", "-c", src, "-o", o]))
        tobjs[tobjs.index(os.path.join(OBJ, base + ".o"))] = o
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
            def one(job):
                if verbose:
                    print(job[0], flush=True)
                _run(job[1])
            list(pool.map(one, jobs))                # re-raises the first failure
    out = os.path.join(LIB, "libsift3d_amd.so")
    if any(_newer(o, out) for o in objs):
        # -Bsymbolic: the library's own calls to init_im & co bind to itself even if another libimutil
        # (e.g. the reference oracle in a test process) is loaded.
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", out, *objs, "-lm", "-lz",
              "-lpthread", "-ldl"])
    # the TESTING variant: the same objects, TESTING_SOURCES recompiled with -DS3D_TESTING (above)
    tout = os.path.join(LIB, "libsift3d_amd_testing.so")
    if any(_newer(o, tout) for o in tobjs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", tout, *tobjs, "-lm", "-lz",
              "-lpthread", "-ldl"])
    synth = os.path.join(LIB, "libs3d_synth.so")
    ssrc = os.path.join(CSRC, "synth.c")
    if _newer(ssrc, synth):
        _run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", synth, ssrc, "-lm"])
    # command-line programs with the reference's argument surface (cli/*.c), linked against the library
    os.makedirs(BIN, exist_ok=True)
    for prog in CLI_PROGRAMS:
        src = os.path.join(ROOT, "cli", prog + ".c")
        exe = os.path.join(BIN, prog)
        if _newer(src, exe, (out, os.path.join(INC, "sift3d_amd.h"))):
            _run(["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", f"-I{INC}", "-o", exe, src, f"-L{LIB}", "-lsift3d_amd",
                  "-lm", "-Wl,-rpath,$ORIGIN/../lib"])
    return out


if __name__ == "__main__":
    print(build(verbose=True))
