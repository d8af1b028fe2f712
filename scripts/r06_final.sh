#!/bin/bash
# round 6, closing visit: the whole GPU suite, both fuzz modes, the PMC traffic of the fused Gaussian (ISA-keyed
# profiles/pmc_gauss.json, so that the bench line of record carries roofline.traffic), the bench line, and two kernel-stat
# summaries: the bench itself (kernels of octaves >= 1 run beside octave 0's on their own streams) and the Gaussian kernels
# alone (scripts/gauss_only.py: nothing beside them) -- the roofline can be recomputed from either table.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
if [ -z "$SKIP_TESTS" ]; then
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/r06_final_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r06_final_pytest.log )
tail -n 4 gpurun_out/r06_final_pytest.log
( timeout 300 python scripts/fuzz_parity.py ${FUZZ_S:-100} ${FUZZ_SEED:-71} > gpurun_out/r06_fuzz.log 2>&1; echo "fuzz exit $?" >> gpurun_out/r06_fuzz.log )
tail -n 2 gpurun_out/r06_fuzz.log
( timeout 300 python scripts/fuzz_parity.py ${FUZZ_S:-100} ${FUZZ_SEED:-72} nonfinite > gpurun_out/r06_fuzz_nonfinite.log 2>&1; echo "fuzz exit $?" >> gpurun_out/r06_fuzz_nonfinite.log )
tail -n 2 gpurun_out/r06_fuzz_nonfinite.log
fi
if [ -z "$SKIP_PMC" ]; then      # (profiles/pmc_gauss.json is keyed on the kernels' machine code: only s3d_gauss.hip edits need a new one)
COMMIT=${COMMIT:-unknown} bash scripts/pmc_hbm.sh r06 > gpurun_out/r06_pmc.log 2>&1
cp gpurun_out/r06_pmc_gauss.json profiles/pmc_gauss.json 2>/dev/null        # (this copy of the repo only: bench.py below reads it)
fi
( timeout 900 python bench.py --steps 20 --warmup 2 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; echo "bench exit $?" >> gpurun_out/r06_final_bench.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_final_bench.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"))
print("roofline frac", r["frac"], "traffic", r.get("traffic"), "physical_frac", r.get("physical_frac"), "|", r.get("traffic_source"))
for k in ("aniso_0.7x0.7x1.5","odd_511","dense_256","two_volume_match"): print(k,c.get(k))
print(c["describe_kernel"]["windows_described_twice"], d.get("cpu_baseline"))
PY
tail -n 2 gpurun_out/r06_final_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/r06_bench_prof.json" 2> "$R/gpurun_out/r06_bench_prof.err"; echo "prof exit $?" >> "$R/gpurun_out/r06_bench_prof.err" )
f=$(find gpurun_out/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/r06_final_kernel_stats.md
head -n 10 gpurun_out/r06_final_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o g -- python "$R/scripts/gauss_only.py" > "$R/gpurun_out/r06_gauss_alone.log" 2>&1 )
f=$(find gpurun_out/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/r06_gauss_alone_kernel_stats.md
grep gauss gpurun_out/r06_gauss_alone_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o d -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/r06_dense_alone.log" 2>&1 )
f=$(find gpurun_out/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/r06_dense_kernel_stats.md
grep -E "bary|dmarch|gauss|scale" gpurun_out/r06_dense_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof
find gpurun_out -name "*.db" -delete
