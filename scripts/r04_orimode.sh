#!/bin/bash
# orientation window tables on unit-voxel volumes: S3D_ORI_MODE 0 (default there) / 1 / 2, testing build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
: > gpurun_out/orimode.txt
for rep in 1 2; do for m in 0 1 2; do
  echo -n "mode $m: " >> gpurun_out/orimode.txt
  SIFT3D_AMD_LIB=$R/sift3d_amd/lib/libsift3d_amd_testing.so S3D_ORI_MODE=$m REPS=8 timeout 200 python scripts/detect_one.py >> gpurun_out/orimode.txt 2>&1
done; done
cat gpurun_out/orimode.txt
