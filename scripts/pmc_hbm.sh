#!/bin/bash
# GPU box: HBM traffic per kernel from the TCC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes, --kernel-trace only.
#   (1) scripts/gauss_only.py  (the six fused Gaussians of the default bank at 512^3)  -> gpurun_out/<tag>_pmc_gauss.json/.md
#   (2) scripts/describe_only.py (one detect + three describes at 512^3: every kernel of a step) -> gpurun_out/<tag>_pmc_hbm_step.md
# usage: COMMIT=<git rev> scripts/pmc_hbm.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for W in gauss_only describe_only; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d "$R/gpurun_out/${TAG}_${W}_$CNT" -o pmc -- python "$R/scripts/$W.py" > "$R/gpurun_out/${TAG}_${W}_$CNT.log" 2>&1
  done
done
python "$R/scripts/pmc_hbm_json.py" "$TAG" "${COMMIT:-unknown}"
find "$R/gpurun_out" -name "*.db" -size +30M -delete
