#!/bin/bash
# time library variants (sift3d_amd/lib/ablate/libsift3d_amd_g<name>.so) through bench.py: detect / describe per step
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
: > gpurun_out/variants.txt
for v in ${VARIANTS}; do
  SIFT3D_AMD_LIB=$R/sift3d_amd/lib/ablate/libsift3d_amd_g$v.so timeout 300 python bench.py --steps ${STEPS:-6} --warmup 1 --no-match --no-cpu-baseline --no-roofline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python - "$v" <<'PY' >> gpurun_out/variants.txt
import json,sys
try:
    d=json.loads(open("gpurun_out/var_%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1],"ms_per_step",d["ms_per_step"],"detect",c.get("detect_ms"),"describe",c.get("describe_ms"),"keypoints",c.get("keypoints"))
except Exception as e: print(sys.argv[1],"failed",e, open("gpurun_out/var_%s.err"%sys.argv[1]).read()[-300:])
PY
done
cat gpurun_out/variants.txt
