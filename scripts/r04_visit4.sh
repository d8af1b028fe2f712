#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "extrema or detect or golden" > gpurun_out/v4_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/v4_tests.log
tail -n 4 gpurun_out/v4_tests.log
run() {  # tag dims units
  ( cd /tmp && export TMPDIR=/tmp && DIMS=$2 UNITS=$3 REPS=4 timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$1" -o t -- python "$R/scripts/detect_one.py" > "$R/gpurun_out/tr_$1.log" 2>&1 )
  f=$(find gpurun_out/tr_$1 -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/trace_$1.md
  tail -n 1 gpurun_out/tr_$1.log; head -n 14 gpurun_out/trace_$1.md
  rm -rf gpurun_out/tr_$1
}
run unit512 512,512,512 1,1,1
SIFT3D_AMD_LIB=$R/sift3d_amd/lib/libsift3d_amd_testing.so S3D_NO_EXTREMA_OVERLAP=1 run unit512_serial 512,512,512 1,1,1
timeout 600 python scripts/tab_time.py detects > gpurun_out/v4_tab_detects.txt 2>&1; echo "exit $?" >> gpurun_out/v4_tab_detects.txt
grep "mode  0" gpurun_out/v4_tab_detects.txt
