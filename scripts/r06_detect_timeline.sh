R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for v in noearly; do
  if [ $v = noearly ]; then export S3D_NO_EARLY_EXTREMA=1; else unset S3D_NO_EARLY_EXTREMA; fi
  SIFT3D_AMD_LIB=$R/sift3d_amd/lib/libsift3d_amd_testing.so REPS=3 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$v -o t -- python $R/scripts/detect_one.py > $R/gpurun_out/tl_$v.log 2>&1
  f=$(find $R/gpurun_out/tl_$v -name "*.db" | head -1)
  python $R/scripts/trace_timeline.py $f k_absmax > $R/gpurun_out/r06_detect_timeline_$v.md
  rm -rf $R/gpurun_out/tl_$v
done

echo ======

