#!/bin/bash
# BASELINE configs[3] on one GPU: the 1024^3 volume whole, and as eight loop-back Z-slab ranks sharing the GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 600 python bench.py --size 1024 --steps 2 --warmup 1 --no-match --no-cpu-baseline --no-roofline > gpurun_out/bench_single_1024.json 2> gpurun_out/bench_single_1024.err; echo "exit $?" >> gpurun_out/bench_single_1024.err )
head -c 900 gpurun_out/bench_single_1024.json; echo; tail -n 2 gpurun_out/bench_single_1024.err
( timeout 900 python bench.py --loopback 8 --steps 2 --warmup 1 --no-match --no-cpu-baseline --no-roofline > gpurun_out/bench_loopback8_strong.json 2> gpurun_out/bench_loopback8_strong.err; echo "exit $?" >> gpurun_out/bench_loopback8_strong.err )
head -c 1500 gpurun_out/bench_loopback8_strong.json; echo; tail -n 2 gpurun_out/bench_loopback8_strong.err
