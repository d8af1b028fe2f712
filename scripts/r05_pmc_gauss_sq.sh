#!/bin/bash
# GPU box: SQ counters of the fused Gaussian kernels (scripts/gauss_only.py, REPS launches per width), separate --pmc passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 REPS=8
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE"
i=0; dbs=""
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d "$R/gpurun_out/${TAG}_gsq$i" -o pmc -- python "$R/scripts/gauss_only.py" > "$R/gpurun_out/${TAG}_gsq$i.log" 2>&1
  f=$(find "$R/gpurun_out/${TAG}_gsq$i" -name "*.db" | head -1); [ -n "$f" ] && dbs="$dbs $f"
done
python "$R/scripts/pmc_summary.py" $dbs > "$R/gpurun_out/${TAG}_pmc_gauss_sq.md" 2>&1
cut -c1-260 "$R/gpurun_out/${TAG}_pmc_gauss_sq.md"
find "$R/gpurun_out" -name "*.db" -size +30M -delete; rm -rf "$R/gpurun_out/${TAG}_gsq1" "$R/gpurun_out/${TAG}_gsq2"
