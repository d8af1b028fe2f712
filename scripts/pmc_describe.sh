#!/bin/bash
# GPU box: SQ counters of the descriptor kernel (scripts/describe_only.py), separate rocprofv3 --pmc passes.
# usage: scripts/pmc_describe.sh <tag>   -> gpurun_out/<tag>_pmc_describe.md
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAVES"
P3="SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"
i=0; dbs=""
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d "$R/gpurun_out/${TAG}_pmc$i" -o pmc -- python "$R/scripts/describe_only.py" > "$R/gpurun_out/${TAG}_pmc$i.log" 2>&1
  f=$(find "$R/gpurun_out/${TAG}_pmc$i" -name "*.db" | head -1); [ -n "$f" ] && dbs="$dbs $f"
done
python "$R/scripts/pmc_summary.py" $dbs > "$R/gpurun_out/${TAG}_pmc_describe.md" 2>&1
cat "$R/gpurun_out/${TAG}_pmc_describe.md"
find "$R/gpurun_out" -name "*.db" -size +30M -delete
