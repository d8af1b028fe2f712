#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=dense
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
i=0; dbs=""
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d "$R/gpurun_out/${TAG}_pmc$i" -o pmc -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/${TAG}_pmc$i.log" 2>&1
  f=$(find "$R/gpurun_out/${TAG}_pmc$i" -name "*.db" | head -1); [ -n "$f" ] && dbs="$dbs $f"
done
python "$R/scripts/pmc_summary.py" $dbs > "$R/gpurun_out/${TAG}_pmc.md" 2>&1
rm -rf "$R"/gpurun_out/${TAG}_pmc[0-9]
python - <<PY
rows=[l.split("|") for l in open("$R/gpurun_out/${TAG}_pmc.md") if l.startswith("|")]
hdr=[h.strip() for h in rows[0]]
for r in rows[2:5]:
    print(r[1].strip())
    for h,v in zip(hdr[2:],r[2:]):
        print("   %-24s %s"%(h,v.strip()))
PY
