#!/bin/bash
# round 6: SQ counters and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dense pipeline's kernels (scripts/dense_only.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=r06_dense
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES"
i=0; dbs=""
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d "$R/gpurun_out/${TAG}_pmc$i" -o pmc -- python "$R/scripts/dense_only.py" > "$R/gpurun_out/${TAG}_pmc$i.log" 2>&1
  f=$(find "$R/gpurun_out/${TAG}_pmc$i" -name "*.db" | head -1); [ -n "$f" ] && dbs="$dbs $f"
done
python "$R/scripts/pmc_summary.py" $dbs > "$R/gpurun_out/${TAG}_pmc.md" 2>&1
rm -rf "$R"/gpurun_out/${TAG}_pmc[0-9]
python - <<PY
rows=[l.split("|") for l in open("$R/gpurun_out/${TAG}_pmc.md") if l.startswith("|")]
hdr=[h.strip() for h in rows[0]]
for r in rows[2:6]:
    print(r[1].strip())
    for h,v in zip(hdr[2:],r[2:]):
        print("   %-24s %s"%(h,v.strip()))
PY
