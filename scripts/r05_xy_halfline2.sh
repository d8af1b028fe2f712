#!/bin/bash
# GPU-box: whole detects at 512^3, HEAD ("old") against the half-line k_gauss_xy ("new"), alternated four times.
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3 4; do for v in old new; do
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so timeout 120 python scripts/detect_ab.py
done; done > gpurun_out/r05_xy_halfline_detect_ab.txt 2>&1
cat gpurun_out/r05_xy_halfline_detect_ab.txt
