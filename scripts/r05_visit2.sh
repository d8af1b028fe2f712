#!/bin/bash
# round 5, second GPU visit: the whole GPU suite, both fuzz modes, a bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log )
tail -n 8 gpurun_out/pytest.log
( timeout 300 python scripts/fuzz_parity.py ${FUZZ_S:-120} ${FUZZ_SEED:-51} > gpurun_out/fuzz.log 2>&1; echo "fuzz exit $?" >> gpurun_out/fuzz.log )
tail -n 3 gpurun_out/fuzz.log
( timeout 300 python scripts/fuzz_parity.py ${FUZZ_S:-120} ${FUZZ_SEED:-52} nonfinite > gpurun_out/fuzz_nonfinite.log 2>&1; echo "fuzz exit $?" >> gpurun_out/fuzz_nonfinite.log )
tail -n 3 gpurun_out/fuzz_nonfinite.log
( timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("value",d["value"],"ms",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"), "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source"))
for k in ("aniso_0.7x0.7x1.5","odd_511","dense_256","two_volume_match"): print(k,c.get(k))
print(c["describe_kernel"]["windows_described_twice"])
PY
tail -n 3 gpurun_out/bench.err
