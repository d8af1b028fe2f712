#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/gauss_round.txt
: > $out
for v in shipped gr2 glds shipped; do
  lib=$R/sift3d_amd/lib/ablate/libsift3d_amd_$v.so
  [ "$v" = "shipped" ] && lib=$R/sift3d_amd/lib/libsift3d_amd.so
  echo "== $v" >> $out
  SIFT3D_AMD_LIB=$lib timeout 200 python scripts/gauss_time.py >> $out 2>&1
done
echo "== parity (shipped)" >> $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "sep_fir or benchmark_size" 2>&1 | tail -n 3 >> $out
cat $out
