#!/bin/bash
# round 5: kernel stats of a bench run + the PMC HBM-traffic passes (profiles/pmc_gauss.json is keyed on the machine code of the
# fused Gaussian kernels: sift3d_amd/codeobj.py), then a bench line that picks the fresh figure up.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/bench_prof.json" 2> "$R/gpurun_out/bench_prof.err"; echo "prof exit $?" >> "$R/gpurun_out/bench_prof.err" )
f=$(find gpurun_out/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/kernel_stats.md
head -n 16 gpurun_out/kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof
COMMIT=${COMMIT:-unknown} bash scripts/pmc_hbm.sh r05 > gpurun_out/pmc.log 2>&1
head -n 12 gpurun_out/r05_pmc_gauss.md | cut -c1-200
cp gpurun_out/r05_pmc_gauss.json profiles/pmc_gauss.json
( timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench exit $?" >> gpurun_out/bench2.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench2.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"]); print(json.dumps(d["roofline"], indent=0)[:1500])
PY
