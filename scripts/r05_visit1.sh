#!/bin/bash
# round 5, first GPU visit: the new tests (non-finite voxels, full-size any-spacing goldens, proof lanes), then a bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -k "nonfinite or seqmax or any_spacing_full_size or proof or redo or runmax or benchmark_size" > gpurun_out/pytest_new.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_new.log )
tail -n 25 gpurun_out/pytest_new.log
( timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("value",d["value"],"ms",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"), "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("physical_frac"))
for k in ("aniso_0.7x0.7x1.5","odd_511","dense_256","two_volume_match"): print(k,c.get(k))
print(c["describe_kernel"]["windows_described_twice"])
PY
tail -n 3 gpurun_out/bench.err
