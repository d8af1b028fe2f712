#!/bin/bash
# GPU-box helper (round 3): the tile kernel for small octaves and the orientation window tables -- parity tests, detect
# timing per variant in one process, a kernel trace of the default configuration, a short bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/detect_round${TAG:-7}.txt
: > $out
echo "== parity" >> $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "${PARITY_K:-tile3 or orient_tables or golden or detect_describe or sep_fir_slab}" 2>&1 | tail -n 6 >> $out
timeout 600 python -m pytest tests/test_gpu_slab.py -q -x -p no:cacheprovider -k "${SLAB_K:-not 1024}" 2>&1 | tail -n 4 >> $out
echo "== variants" >> $out
timeout 600 python scripts/detect_variants.py >> $out 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_detect${TAG:-7}" -o detect -- python "$R/scripts/detect_ab.py" > "$R/gpurun_out/prof_detect${TAG:-7}.log" 2>&1 )
if [ -n "$DO_UBENCH" ]; then
  echo "== MFMA micro-benchmark" >> $out
  ( timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma.hip -o /tmp/um > /dev/null 2>&1 && timeout 60 /tmp/um ) >> $out 2>&1
fi
echo "== bench" >> $out
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick${TAG:-7}.json 2>> $out
cut -c1-700 gpurun_out/bench_quick${TAG:-7}.json >> $out
cat $out
