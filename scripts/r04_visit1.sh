#!/bin/bash
# Round-4 visit 1: parity of the table-driven passes on the GPU, then their timing against the kernels they replace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sep_fir_tab or detect_full_size_any_spacing or sep_fir_slab or sep_fir_api" > gpurun_out/v1_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/v1_tests.log
tail -n 15 gpurun_out/v1_tests.log
REPS=10 timeout 600 python scripts/tab_time.py passes > gpurun_out/v1_tab_passes.txt 2>&1; echo "exit $?" >> gpurun_out/v1_tab_passes.txt
cat gpurun_out/v1_tab_passes.txt
timeout 600 python scripts/tab_time.py detects > gpurun_out/v1_tab_detects.txt 2>&1; echo "exit $?" >> gpurun_out/v1_tab_detects.txt
cat gpurun_out/v1_tab_detects.txt
