#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
: > gpurun_out/ext_ablate.txt
for v in ${VARIANTS:-u2 u3 u4 u2 u3}; do
  ( cd /tmp && export TMPDIR=/tmp && SIFT3D_AMD_LIB=$R/sift3d_amd/lib/ablate/libsift3d_amd_g$v.so S3D_NO_EXTREMA_OVERLAP=1 REPS=4 timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$v" -o t -- python "$R/scripts/detect_one.py" > "$R/gpurun_out/tr_$v.log" 2>&1 )
  f=$(find gpurun_out/tr_$v -name "*.db" | head -1)
  echo "== $v: $(tail -n 1 gpurun_out/tr_$v.log)" >> gpurun_out/ext_ablate.txt
  [ -n "$f" ] && python scripts/prof_summary.py $f | grep "k_extrema" >> gpurun_out/ext_ablate.txt
  rm -rf gpurun_out/tr_$v
done
cat gpurun_out/ext_ablate.txt
