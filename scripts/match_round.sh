#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/match_round.txt
: > $out
for v in ${VARIANTS_M:-shipped gmold shipped gmold}; do
  lib=$R/sift3d_amd/lib/ablate/libsift3d_amd_$v.so
  [ "$v" = "shipped" ] && lib=$R/sift3d_amd/lib/libsift3d_amd.so
  echo "== $v" >> $out
  SIFT3D_AMD_LIB=$lib timeout 200 python scripts/match_ab.py >> $out 2>&1
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_match" -o match -- python "$R/scripts/match_ab.py" > "$R/gpurun_out/prof_match.log" 2>&1 )
echo "== parity (shipped)" >> $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "nn_match or two_volume" 2>&1 | tail -n 3 >> $out
cat $out
