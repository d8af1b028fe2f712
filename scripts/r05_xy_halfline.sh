#!/bin/bash
# GPU-box: A/B of k_gauss_xy's LDS line layout (HEAD: float4 line, ds_read_b128; WORK: two half lines, ds_read2_b64): HIP-event
# times per width at 512^3, alternated twice, then whole detects.
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in old new old new; do
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so REPS=20 timeout 300 python scripts/gauss_time.py
done > gpurun_out/r05_xy_halfline_ab.txt 2>&1
for v in old new; do
  echo "## detects, variant $v"
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so MODES=0 timeout 300 python scripts/tab_time.py detects
done >> gpurun_out/r05_xy_halfline_ab.txt 2>&1
cat gpurun_out/r05_xy_halfline_ab.txt
