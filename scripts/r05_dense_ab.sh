#!/bin/bash
# round 5: the dense x pass, A/B of builds under sift3d_amd/lib/ablate (scripts/build_file_variants.py s3d_gauss ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense" > gpurun_out/dense_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/dense_tests.log
tail -n 3 gpurun_out/dense_tests.log
for so in "" $(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so 2>/dev/null); do
  tag=$(basename "${so:-default}" .so)
  ( cd /tmp && export TMPDIR=/tmp && SIFT3D_AMD_LIB=${so:+$R/$so} timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$tag" -o t -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/tr_$tag.log" 2>&1 )
  f=$(find gpurun_out/tr_$tag -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/trace_dense_$tag.md
  echo "== $tag"; tail -n 2 gpurun_out/tr_$tag.log; grep -E "bary|march|dense_post" gpurun_out/trace_dense_$tag.md | cut -c1-160
  rm -rf gpurun_out/tr_$tag
done
