# round 6: kernel by kernel, what a 512^3 volume with one NaN voxel costs (scripts/nonfinite_cost.py under rocprofv3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
rm -rf gpurun_out/prof
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o x -- python "$R/scripts/nonfinite_cost.py" 2>&1 | grep "detect" )
f=$(find gpurun_out/prof -name "*.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/r06_nonfinite_kernel_stats.md
rm -rf gpurun_out/prof
