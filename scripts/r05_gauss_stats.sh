#!/bin/bash
# kernel stats of the fused Gaussians ALONE (scripts/gauss_only.py: the six widths of the default bank at 512^3, nothing beside them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/gs" -o g -- python "$R/scripts/gauss_only.py" > "$R/gpurun_out/gs.log" 2>&1 )
f=$(find gpurun_out/gs -name "*.db" | head -1); [ -n "$f" ] && python scripts/prof_summary.py $f > gpurun_out/gauss_only_kernel_stats.md
cat gpurun_out/gauss_only_kernel_stats.md | cut -c1-140; rm -rf gpurun_out/gs
