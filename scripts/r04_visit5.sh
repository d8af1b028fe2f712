#!/bin/bash
# orientation tables for any spacing (tests + bench extras) and the descriptor back-end timing variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "orient or aniso or any_spacing or kpSift3D or two_volume or detect_describe" > gpurun_out/pytest_v5.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_v5.log )
tail -n 8 gpurun_out/pytest_v5.log
( timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/bench_v5.json 2> gpurun_out/bench_v5.err; echo "bench exit $?" >> gpurun_out/bench_v5.err )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_v5.json").read().strip().splitlines()[-1])
c=d["config"]
print("value",d["value"],"ms",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"))
for k in ("aniso_0.7x0.7x1.5","odd_511"): print(k,c.get(k))
PY
: > gpurun_out/describe_variants.txt
for v in dbase df32 du32 dbase df32 du32; do
  SIFT3D_AMD_LIB=$R/sift3d_amd/lib/ablate/libsift3d_amd_g$v.so timeout 300 python bench.py --steps 6 --warmup 1 --no-match --no-cpu-baseline --no-roofline > gpurun_out/dv_$v.json 2> gpurun_out/dv_$v.err
  python - "$v" <<'PY' >> gpurun_out/describe_variants.txt
import json,sys
try:
    d=json.loads(open("gpurun_out/dv_%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1],"ms_per_step",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"))
except Exception as e: print(sys.argv[1],"failed",e)
PY
done
cat gpurun_out/describe_variants.txt
