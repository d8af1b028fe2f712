#!/bin/bash
# GPU-box: A/B of k_conv_march_tab's ring layout (HEAD: float4 rows read by ds_read_b128; WORK: two float2 half rings read by
# ds_read_b64) on per-pass and whole-detect times, then the parity tests of the table-driven passes.
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in old new old new; do
  echo "## variant $v"
  SIFT3D_AMD_LIB=sift3d_amd/lib/ablate/libsift3d_amd_g$v.so MODES=0 timeout 300 python scripts/tab_time.py passes detects
done > gpurun_out/r05_tab_split_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sep_fir or any_spacing or full_size or aniso or ragged" > gpurun_out/r05_tab_split_pytest.log 2>&1
tail -3 gpurun_out/r05_tab_split_pytest.log
