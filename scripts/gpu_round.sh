#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprof kernel stats.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== nproc $(nproc) ; $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' )" > gpurun_out/info.txt
( timeout ${T_TESTS:-900} python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log )
tail -n 40 gpurun_out/pytest.log
if [ -z "$SKIP_BENCH" ]; then
  ( timeout ${T_BENCH:-600} python bench.py --steps ${STEPS:-3} --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err )
  cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
fi
if [ -z "$SKIP_PROF" ]; then
  ( cd /tmp && export TMPDIR=/tmp && timeout ${T_PROF:-600} rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/bench_prof.json" 2> "$R/gpurun_out/bench_prof.err"; echo "prof exit $?" >> "$R/gpurun_out/bench_prof.err" )
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 30 "$f"
  if [ -n "$DO_PMC" ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$R/gpurun_out/pmc_$c" -o pmc -- python "$R/scripts/gauss_only.py" > "$R/gpurun_out/pmc_$c.log" 2>&1; echo "pmc $c exit $?" >> "$R/gpurun_out/pmc_$c.log" )
      tail -n 2 "$R/gpurun_out/pmc_$c.log"
    done
  fi
  # keep the merge small: drop the raw per-dispatch trace
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
