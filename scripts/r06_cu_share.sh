R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T=$R/sift3d_amd/lib/libsift3d_amd_testing.so
for sh in "" "7,8" "3,4" "15,16" "" "7,8"; do
  echo "share=[$sh]"; SIFT3D_AMD_LIB=$T S3D_EXT_CU_SHARE=$sh REPS=8 python scripts/detect_one.py 2>&1 | tail -n 1
done
