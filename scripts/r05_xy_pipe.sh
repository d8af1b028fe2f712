#!/bin/bash
# GPU-box: k_gauss_xy software-pipelined by one row ("pipe"), pipelined + half-line LDS layout ("both"), against HEAD ("old"):
# HIP-event times of the fused kernels per width at 512^3, and whole detects, alternated.
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do for v in old pipe both; do
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so REPS=20 timeout 300 python scripts/gauss_time.py
done; done > gpurun_out/r05_xy_pipe_ab.txt 2>&1
echo "## whole detects at 512^3 (scripts/detect_ab.py), alternated" >> gpurun_out/r05_xy_pipe_ab.txt
for i in 1 2 3; do for v in old pipe both; do
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so timeout 120 python scripts/detect_ab.py
done; done >> gpurun_out/r05_xy_pipe_ab.txt 2>&1
cat gpurun_out/r05_xy_pipe_ab.txt
