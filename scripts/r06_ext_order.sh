R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T=$R/sift3d_amd/lib/libsift3d_amd_testing.so
for cf in 0 1 0 1; do
  echo "chains_first=$cf"; SIFT3D_AMD_LIB=$T S3D_EXT_CHAINS_FIRST=$cf REPS=8 python scripts/detect_one.py 2>&1 | tail -n 1
done
echo "chains_first=1 without early bitmaps"; SIFT3D_AMD_LIB=$T S3D_EXT_CHAINS_FIRST=1 S3D_NO_EARLY_EXTREMA=1 REPS=8 python scripts/detect_one.py 2>&1 | tail -n 1
for cf in 0 1; do echo "aniso chains_first=$cf"; SIFT3D_AMD_LIB=$T S3D_EXT_CHAINS_FIRST=$cf REPS=6 DIMS=512,512,300 UNITS=0.7,0.7,1.5 python scripts/detect_one.py 2>&1 | tail -n 1; done
for cf in 0 1; do echo "256 chains_first=$cf"; SIFT3D_AMD_LIB=$T S3D_EXT_CHAINS_FIRST=$cf REPS=6 DIMS=256,256,256 python scripts/detect_one.py 2>&1 | tail -n 1; done
