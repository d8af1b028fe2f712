#!/bin/bash
# round 5: coarse octaves on streams of their own (build_gpyr_dev): A/B through the TESTING library's S3D_NO_OCTAVE_STREAMS
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=$R/sift3d_amd/lib/libsift3d_amd_testing.so
for rep in 1 2 3; do
  SIFT3D_AMD_LIB=$T S3D_NO_OCTAVE_STREAMS=1 python scripts/detect_ab.py 2>&1 | tail -1 | sed 's/^/one stream : /'
  SIFT3D_AMD_LIB=$T python scripts/detect_ab.py 2>&1 | tail -1 | sed 's/^/octave streams: /'
done | tee gpurun_out/octstreams.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "detect or benchmark_size or two_volume or struct_reuse or two_sift3d" > gpurun_out/oct_tests.log 2>&1; tail -n 3 gpurun_out/oct_tests.log
