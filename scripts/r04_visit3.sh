#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sep_fir_tab or sep_fir_div" > gpurun_out/v3_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/v3_tests.log
tail -n 4 gpurun_out/v3_tests.log
REPS=10 timeout 600 python scripts/tab_time.py passes > gpurun_out/v3_tab_passes.txt 2>&1; echo "exit $?" >> gpurun_out/v3_tab_passes.txt
head -n 18 gpurun_out/v3_tab_passes.txt
timeout 600 python scripts/tab_time.py detects > gpurun_out/v3_tab_detects.txt 2>&1; echo "exit $?" >> gpurun_out/v3_tab_detects.txt
grep "mode  0" gpurun_out/v3_tab_detects.txt
