#!/bin/bash
# round 6: k_extrema_fused workgroup size (-DEXF_BLOCK=64/128/256/512 builds from scripts/build_file_variants.py s3d_extrema ...):
# detect wall time at 512^3 (keypoint count must not move) and the kernel's own duration under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
for rnd in 1 2; do
for so in $(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so); do
  echo "== round $rnd $(basename $so .so)"
  SIFT3D_AMD_LIB=$R/$so REPS=12 timeout 300 python scripts/detect_one.py 2>&1 | tail -n 1
done; done
for so in $(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so); do
  echo "== rocprof $(basename $so .so)"
  rm -rf gpurun_out/prof
  ( cd /tmp && export TMPDIR=/tmp && SIFT3D_AMD_LIB=$R/$so REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o x -- python "$R/scripts/detect_one.py" > /dev/null 2>&1 )
  f=$(find gpurun_out/prof -name "*.db" | head -1); python scripts/prof_summary.py $f | grep -E "k_extrema_fused|k_gauss_z<2" | cut -c1-200
done
rm -rf gpurun_out/prof
