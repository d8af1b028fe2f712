#!/bin/bash
# SQ counters of every kernel of a detect (separate rocprofv3 --pmc passes).  usage: r04_pmc_detect.sh <tag> <dims> <units>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-unit512}; DIMS=${2:-512,512,512}; UNITS=${3:-1,1,1}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"
i=0; dbs=""
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  SIFT3D_AMD_LIB=$R/sift3d_amd/lib/libsift3d_amd_testing.so S3D_NO_EXTREMA_OVERLAP=1 DIMS=$DIMS UNITS=$UNITS REPS=2 timeout 300 rocprofv3 --pmc $P --kernel-trace -d "$R/gpurun_out/${TAG}_pmc$i" -o pmc -- python "$R/scripts/detect_one.py" > "$R/gpurun_out/${TAG}_pmc$i.log" 2>&1
  f=$(find "$R/gpurun_out/${TAG}_pmc$i" -name "*.db" | head -1); [ -n "$f" ] && dbs="$dbs $f"
done
python "$R/scripts/pmc_summary.py" $dbs > "$R/gpurun_out/${TAG}_pmc_detect.md" 2>&1
cat "$R/gpurun_out/${TAG}_pmc_detect.md"
rm -rf "$R"/gpurun_out/${TAG}_pmc[0-9]
