#!/bin/bash
# GPU-box: A/B of k_nn_gemm's LDS panel layout (HEAD: row-major pitch 40 halves, ds_read_b128 fragments; WORK: k-group major,
# ds_read2_b64), match of 31207 x 31207 descriptors, alternated; then the matcher's parity tests on the new build.
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in old new old new; do
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so timeout 300 python scripts/match_ab.py
done > gpurun_out/r05_gemm_lds_ab.txt 2>&1
cat gpurun_out/r05_gemm_lds_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "match or nn" > gpurun_out/r05_gemm_lds_pytest.log 2>&1
tail -3 gpurun_out/r05_gemm_lds_pytest.log
