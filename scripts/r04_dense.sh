#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense" > gpurun_out/dense_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/dense_tests.log
tail -n 3 gpurun_out/dense_tests.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_dense" -o t -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/tr_dense.log" 2>&1 )
f=$(find gpurun_out/tr_dense -name "*.db" | head -1)
[ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/trace_dense.md
tail -n 2 gpurun_out/tr_dense.log; head -n 14 gpurun_out/trace_dense.md
rm -rf gpurun_out/tr_dense
