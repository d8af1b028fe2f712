#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
: > gpurun_out/dense_ablate.txt
for v in ${VARIANTS:-pd2 pd3 pd4 pd6}; do
  ( cd /tmp && export TMPDIR=/tmp && SIFT3D_AMD_LIB=$R/sift3d_amd/lib/ablate/libsift3d_amd_g$v.so timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$v" -o t -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/tr_$v.log" 2>&1 )
  f=$(find gpurun_out/tr_$v -name "*.db" | head -1)
  echo "== $v: $(grep dense gpurun_out/tr_$v.log | tail -n 1)" >> gpurun_out/dense_ablate.txt
  [ -n "$f" ] && python scripts/prof_summary.py $f | grep "k_march\|k_bary\|k_dense" >> gpurun_out/dense_ablate.txt
  rm -rf gpurun_out/tr_$v
done
cat gpurun_out/dense_ablate.txt
