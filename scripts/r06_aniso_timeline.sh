# round 6: kernel timeline of one detect of the 0.7 x 0.7 x 1.5 mm volume (every Gaussian pass table-driven)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
DIMS=512,512,300 UNITS=0.7,0.7,1.5 REPS=3 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o t -- python $R/scripts/detect_one.py > $R/gpurun_out/tl.log 2>&1
f=$(find $R/gpurun_out/tl -name "*.db" | head -1)
python $R/scripts/trace_timeline.py $f k_absmax > $R/gpurun_out/r06_aniso_timeline.md
rm -rf $R/gpurun_out/tl
