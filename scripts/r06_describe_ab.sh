#!/bin/bash
# round 6: descriptor-kernel build variants (sift3d_amd/lib/ablate/libsift3d_amd_g*.so, scripts/build_file_variants.py s3d_keypoint ...)
# timed against each other on the 512^3 bench volume; descriptors compared with the first one (tolerance units)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
first=""
for so in ${DWV:-$(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so)}; do
  echo "== $(basename $so .so)"
  if [ -z "$first" ]; then
    SIFT3D_AMD_LIB=$R/$so SAVE=/tmp/desc_first.npy timeout 300 python scripts/describe_ab.py 2>&1 | tail -n 3
    first=$so
  else
    SIFT3D_AMD_LIB=$R/$so CMP=/tmp/desc_first.npy timeout 300 python scripts/describe_ab.py 2>&1 | tail -n 3
  fi
done
