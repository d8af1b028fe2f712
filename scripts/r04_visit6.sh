#!/bin/bash
# 32-bit histogram fields in the descriptor kernel: parity subset + bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "${KEXPR:-describe or detect or two_volume or kpSift3D or reg or dense}" > gpurun_out/pytest_v6.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_v6.log )
tail -n 8 gpurun_out/pytest_v6.log
( timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/bench_v6.json 2> gpurun_out/bench_v6.err; echo "bench exit $?" >> gpurun_out/bench_v6.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_v6.json").read().strip().splitlines()[-1])
c=d["config"]
print("value",d["value"],"ms",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"))
print(c.get("describe_kernel"))
for k in ("aniso_0.7x0.7x1.5","odd_511","two_volume_512"): print(k,c.get(k))
PY
tail -n 3 gpurun_out/bench_v6.err
cat gpurun_out/golden_pair512*.json 2>/dev/null
