#!/bin/bash
# round 6: table-driven x pass, build variants (scripts/build_file_variants.py s3d_gauss_tab ...): detect wall time of the
# 512 x 512 x 300 volume of 0.7 x 0.7 x 1.5 mm voxels (keypoint count must not move) and the x-pass kernels under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
export DIMS=${DIMS:-512,512,300} UNITS=${UNITS:-0.7,0.7,1.5}
for rnd in 1 2; do
for so in $(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so); do
  echo "== round $rnd $(basename $so .so)"
  SIFT3D_AMD_LIB=$R/$so REPS=12 timeout 300 python scripts/detect_one.py 2>&1 | tail -n 1
done; done
for so in $(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so); do
  echo "== rocprof $(basename $so .so)"
  rm -rf gpurun_out/prof
  ( cd /tmp && export TMPDIR=/tmp && SIFT3D_AMD_LIB=$R/$so REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o x -- python "$R/scripts/detect_one.py" > /dev/null 2>&1 )
  f=$(find gpurun_out/prof -name "*.db" | head -1); python scripts/prof_summary.py $f | grep -E "k_conv_x_tab|k_conv_march_tab" | cut -c1-200
done
rm -rf gpurun_out/prof
