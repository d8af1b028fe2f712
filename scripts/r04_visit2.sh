#!/bin/bash
# Round-4 visit 2: GPU parity of the Gaussian / extrema / image-op / detect tests, then detect timings and traces.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_image_ops.py -x -q -m gpu -k "sep_fir or extrema or detect or image_ops or golden" > gpurun_out/v2_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/v2_tests.log
tail -n 8 gpurun_out/v2_tests.log
timeout 600 python scripts/tab_time.py detects > gpurun_out/v2_tab_detects.txt 2>&1; echo "exit $?" >> gpurun_out/v2_tab_detects.txt
cat gpurun_out/v2_tab_detects.txt
bash scripts/r04_trace_detect.sh
