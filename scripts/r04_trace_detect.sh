#!/bin/bash
# kernel traces of whole detects on a ragged and an anisotropic volume -> gpurun_out/trace_<tag>.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
run() {  # tag dims units
  ( cd /tmp && export TMPDIR=/tmp && DIMS=$2 UNITS=$3 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$1" -o t -- python "$R/scripts/detect_one.py" > "$R/gpurun_out/tr_$1.log" 2>&1 )
  f=$(find gpurun_out/tr_$1 -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/trace_$1.md
  tail -n 1 gpurun_out/tr_$1.log; head -n 28 gpurun_out/trace_$1.md
  rm -rf gpurun_out/tr_$1
}
run odd511 511,509,303 1,1,1
run aniso07 512,512,300 0.7,0.7,1.5
