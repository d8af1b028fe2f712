#!/bin/bash
# GPU-box helper: time the descriptor-kernel build variants of scripts/build_ablate.py against each other on the 512^3
# bench volume and compare their descriptors with the first one (tolerance units).  DWV="base bf ..." PARITY="a1 all"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/describe_variants.txt
: > $out
first=""
for v in ${DWV:-base}; do
  lib=$R/sift3d_amd/lib/ablate/libsift3d_amd_a$v.so
  [ "$v" = "shipped" ] && lib=$R/sift3d_amd/lib/libsift3d_amd.so
  echo "== $v" >> $out
  if [ -z "$first" ]; then
    SIFT3D_AMD_LIB=$lib SAVE=/tmp/desc_first.npy timeout 300 python scripts/describe_ab.py >> $out 2>&1
    first=$v
  else
    SIFT3D_AMD_LIB=$lib CMP=/tmp/desc_first.npy timeout 300 python scripts/describe_ab.py >> $out 2>&1
  fi
done
for v in ${PARITY:-}; do
  lib=$R/sift3d_amd/lib/ablate/libsift3d_amd_a$v.so
  echo "== parity $v" >> $out
  SIFT3D_AMD_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "${PARITY_K:-window_set or detect_describe}" 2>&1 | tail -n 5 >> $out
done
cat $out
