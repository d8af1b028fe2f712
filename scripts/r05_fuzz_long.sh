#!/bin/bash
# GPU-box: longer fuzz with seeds of their own (finite shapes / spacings / densities, then non-finite volumes), against the CPU oracle
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( timeout 400 python scripts/fuzz_parity.py ${FUZZ_S:-240} ${SEED_A:-71} > gpurun_out/r05_fuzz_long.log 2>&1; echo "fuzz exit $?" >> gpurun_out/r05_fuzz_long.log )
tail -n 3 gpurun_out/r05_fuzz_long.log
( timeout 400 python scripts/fuzz_parity.py ${FUZZ_S:-240} ${SEED_B:-72} nonfinite > gpurun_out/r05_fuzz_long_nonfinite.log 2>&1; echo "fuzz exit $?" >> gpurun_out/r05_fuzz_long_nonfinite.log )
tail -n 3 gpurun_out/r05_fuzz_long_nonfinite.log
