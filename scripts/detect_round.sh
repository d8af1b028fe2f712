#!/bin/bash
# GPU-box helper: detect timing with and without the octave-chain overlap, the detect/describe goldens, a kernel trace of
# two detects (timeline), and a short bench.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/detect_round.txt
: > $out
echo "== detect, octave chain on its own stream" >> $out
timeout 300 python scripts/detect_ab.py >> $out 2>&1
echo "== detect, S3D_NO_OCTAVE_OVERLAP=1" >> $out
S3D_NO_OCTAVE_OVERLAP=1 timeout 300 python scripts/detect_ab.py >> $out 2>&1
echo "== parity" >> $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "${PARITY_K:-golden or sep_fir or detect_describe or image_ops}" 2>&1 | tail -n 5 >> $out
timeout 300 python -m pytest tests/test_image_ops.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -n 3 >> $out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_detect" -o detect -- python "$R/scripts/detect_ab.py" > "$R/gpurun_out/prof_detect.log" 2>&1 )
echo "== bench" >> $out
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick.json 2>> $out
cat gpurun_out/bench_quick.json | cut -c1-600 >> $out
cat $out
