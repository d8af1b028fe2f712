#!/bin/bash
# Round-3 GPU-box visit: gpu_round.sh (tests, bench, kernel stats), the PMC passes, the N > 1 code paths a one-GPU box can
# run (refusal of --gpus 2, the torchrun form with a world of one: gloo bootstrap + real RCCL), loop-back weak scaling on
# one GPU, and the CPU baseline on the real 512^3 configuration.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
bash scripts/gpu_round.sh
COMMIT=${COMMIT:-unknown} bash scripts/pmc_hbm.sh r03
echo "== bench --gpus 2 on this box (must refuse, exit 2)" > gpurun_out/multi_gpu_paths.txt
python bench.py --gpus 2 --steps 1 --warmup 0 >> gpurun_out/multi_gpu_paths.txt 2>&1; echo "exit $?" >> gpurun_out/multi_gpu_paths.txt
echo "== torchrun form, world of one (gloo bootstrap, RCCL communicators of one rank)" >> gpurun_out/multi_gpu_paths.txt
S3D_BENCH_FORCE_SLAB=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --no-match --no-cpu-baseline --no-roofline >> gpurun_out/multi_gpu_paths.txt 2>&1; echo "exit $?" >> gpurun_out/multi_gpu_paths.txt
echo "== loop-back, 2 ranks on this GPU, 512x512x1024 (decomposition overhead, not scaling)" >> gpurun_out/multi_gpu_paths.txt
timeout 300 python bench.py --loopback 2 --steps 2 --warmup 1 --no-match --no-cpu-baseline --no-roofline >> gpurun_out/multi_gpu_paths.txt 2>&1; echo "exit $?" >> gpurun_out/multi_gpu_paths.txt
echo "== torchrun, two ranks sharing this GPU (gloo bootstrap; callback transport over gloo since two ranks cannot share one GPU over RCCL)" >> gpurun_out/multi_gpu_paths.txt
S3D_BENCH_SAME_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --size 256 --steps 2 --warmup 1 --no-match --no-roofline >> gpurun_out/multi_gpu_paths.txt 2>&1; echo "exit $?" >> gpurun_out/multi_gpu_paths.txt
cat gpurun_out/multi_gpu_paths.txt | cut -c1-400
if [ -n "$DO_FUZZ" ]; then
  ( timeout 600 python scripts/fuzz_parity.py ${FUZZ_SECONDS:-120} ${FUZZ_SEED:-31} > gpurun_out/fuzz_parity.log 2>&1; echo "fuzz exit $?" >> gpurun_out/fuzz_parity.log )
  tail -n 4 gpurun_out/fuzz_parity.log
fi
if [ -n "$DO_CPU512" ]; then
  ( OMP_NUM_THREADS=64 OPENBLAS_NUM_THREADS=1 timeout 900 python bench.py --cpu-baseline-worker 512 > gpurun_out/cpu_baseline_512.json 2> gpurun_out/cpu_baseline_512.err; echo "cpu512 exit $?" >> gpurun_out/cpu_baseline_512.err )
  cat gpurun_out/cpu_baseline_512.json
fi
