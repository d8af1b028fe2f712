R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T=$R/sift3d_amd/lib/libsift3d_amd_testing.so
for sl in 1 8 32 64 128 1 32; do
  echo "slices=$sl"; SIFT3D_AMD_LIB=$T S3D_EXT_SLICES=$sl REPS=8 python scripts/detect_one.py 2>&1 | tail -n 1
done
