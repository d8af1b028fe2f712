#!/bin/bash
# the launch forms of bench.py that one GPU can show: torchrun with a world of one; --gpus 2 on a one-GPU box must refuse
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-match --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "exit $?" >> gpurun_out/bench_torchrun1.err )
head -c 700 gpurun_out/bench_torchrun1.json; echo; tail -n 2 gpurun_out/bench_torchrun1.err
( timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err; echo "exit $?" >> gpurun_out/bench_gpus2.err )
head -c 300 gpurun_out/bench_gpus2.json; echo; tail -n 3 gpurun_out/bench_gpus2.err
( timeout 600 python bench.py --loopback 2 --steps 2 --warmup 1 --strong-size 512 --no-match --no-cpu-baseline --no-roofline > gpurun_out/bench_loopback2.json 2> gpurun_out/bench_loopback2.err; echo "exit $?" >> gpurun_out/bench_loopback2.err )
head -c 900 gpurun_out/bench_loopback2.json; echo; tail -n 2 gpurun_out/bench_loopback2.err
