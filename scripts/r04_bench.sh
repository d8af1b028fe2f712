#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 900 python bench.py --steps ${STEPS:-10} --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err )
cat gpurun_out/bench.json; tail -n 3 gpurun_out/bench.err
if [ -n "$DO_LOOP8" ]; then
  ( timeout 900 python bench.py --loopback 8 --steps 2 --warmup 1 --no-match --no-cpu-baseline --no-roofline > gpurun_out/bench_loopback8_strong.json 2> gpurun_out/bench_loopback8_strong.err; echo "exit $?" >> gpurun_out/bench_loopback8_strong.err )
  cat gpurun_out/bench_loopback8_strong.json; tail -n 2 gpurun_out/bench_loopback8_strong.err
fi
