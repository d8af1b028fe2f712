#!/bin/bash
# round 6: the dense pipeline (k_bary_x_wave, k_dmarch y, k_dmarch z + postproc), A/B of builds under sift3d_amd/lib/ablate
# (scripts/build_file_variants.py s3d_dense ...) and of chunk counts (DENSE_CHUNKS).  usage: r06_dense_ab.sh [chunks ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense" > gpurun_out/dense_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/dense_tests.log
tail -n 3 gpurun_out/dense_tests.log
fi
for so in "" $(ls sift3d_amd/lib/ablate/libsift3d_amd_g*.so 2>/dev/null); do
 for ch in ${@:-0}; do
  tag=$(basename "${so:-default}" .so)_c$ch
  ( cd /tmp && export TMPDIR=/tmp && DENSE_CHUNKS=$ch SIFT3D_AMD_LIB=${so:+$R/$so} timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$tag" -o t -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/tr_$tag.log" 2>&1 )
  f=$(find gpurun_out/tr_$tag -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/trace_dense_$tag.md
  [ -n "$f" ] && [ -n "$TIMELINE" ] && python scripts/trace_timeline.py $f k_gauss_xy > gpurun_out/timeline_dense_$tag.md
  echo "== $tag"; grep dense gpurun_out/tr_$tag.log | tail -n 2; grep -E "bary|march|dense_post|gauss|absmax|scale" gpurun_out/trace_dense_$tag.md | cut -c1-150
  rm -rf gpurun_out/tr_$tag
 done
done
