#!/bin/bash
# full GPU suite + bench + kernel stats of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log )
tail -n 12 gpurun_out/pytest.log
( timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err )
head -c 1200 gpurun_out/bench.json; echo; tail -n 2 gpurun_out/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/bench_prof.json" 2> "$R/gpurun_out/bench_prof.err"; echo "prof exit $?" >> "$R/gpurun_out/bench_prof.err" )
f=$(find gpurun_out/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/kernel_stats.md
head -n 12 gpurun_out/kernel_stats.md
rm -rf gpurun_out/prof
