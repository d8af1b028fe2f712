R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
rm -rf gpurun_out/prof
( cd /tmp && export TMPDIR=/tmp && DIMS=512,512,300 UNITS=0.7,0.7,1.5 REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o x -- python "$R/scripts/detect_one.py" 2>&1 | tail -1 )
f=$(find gpurun_out/prof -name "*.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/r06_aniso_kernel_stats.md
rm -rf gpurun_out/prof
DIMS=512,512,300 UNITS=0.7,0.7,1.5 REPS=12 python scripts/detect_one.py
