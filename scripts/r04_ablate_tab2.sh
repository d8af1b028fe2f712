#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/tab_ablate2.txt; : > $OUT
for v in ${VARIANTS}; do
  for uf in "1/0.7" "1/1.5"; do
    SIFT3D_AMD_LIB=$R/sift3d_amd/lib/ablate/libsift3d_amd_g$v.so UF=$uf AXES=${AXES:-12} WIDTHS=${WIDTHS:-5,9,13,17} DIMS=${DIMS:-512,512,304} timeout 120 python scripts/tab_one.py >> $OUT 2>&1
  done
done
cat $OUT
