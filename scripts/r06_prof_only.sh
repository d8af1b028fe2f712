R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/r06_bench_prof.json" 2> "$R/gpurun_out/r06_bench_prof.err" )
f=$(find gpurun_out/prof -name "*.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/r06_final_kernel_stats.md
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_prof.json").read().strip().splitlines()[-1])
print([ (a["width"], a["xy_ms"], a["z_ms"]) for a in d["config"]["gauss_apps"]])
PY
grep -E "k_gauss_xy<8|k_gauss_z<8|k_gauss_xy<2|k_gauss_z<2|k_describe|k_extrema_fused" gpurun_out/r06_final_kernel_stats.md | cut -c1-170
rm -rf gpurun_out/prof
