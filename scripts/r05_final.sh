#!/bin/bash
# round 5, closing visit: the whole GPU suite, both fuzz modes, the bench line of record (with roofline.traffic from the
# ISA-keyed profiles/pmc_gauss.json), and the kernel stats of that bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log )
tail -n 4 gpurun_out/pytest.log
( timeout 300 python scripts/fuzz_parity.py ${FUZZ_S:-100} ${FUZZ_SEED:-61} > gpurun_out/fuzz.log 2>&1; echo "fuzz exit $?" >> gpurun_out/fuzz.log )
tail -n 2 gpurun_out/fuzz.log
( timeout 300 python scripts/fuzz_parity.py ${FUZZ_S:-100} ${FUZZ_SEED:-62} nonfinite > gpurun_out/fuzz_nonfinite.log 2>&1; echo "fuzz exit $?" >> gpurun_out/fuzz_nonfinite.log )
tail -n 2 gpurun_out/fuzz_nonfinite.log
( timeout 900 python bench.py --steps 20 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"))
print("roofline frac", r["frac"], "traffic", r.get("traffic"), "physical_frac", r.get("physical_frac"), "|", r.get("traffic_source"))
for k in ("aniso_0.7x0.7x1.5","odd_511","dense_256","two_volume_match"): print(k,c.get(k))
print(c["describe_kernel"]["windows_described_twice"], d.get("cpu_baseline"))
PY
tail -n 2 gpurun_out/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/bench_prof.json" 2> "$R/gpurun_out/bench_prof.err"; echo "prof exit $?" >> "$R/gpurun_out/bench_prof.err" )
f=$(find gpurun_out/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/kernel_stats.md
head -n 10 gpurun_out/kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof
