#!/bin/sh
# tests/emu/build_emu.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the product's .hip and host .c sources with plain g++/gcc against the SIMT emulator in
# tests/emu/hip/hip_runtime.h into tests/emu/libsift3d_emu.so, so kernel logic can be parity-tested
# on a GPU-less box.  Never loaded by the sift3d_amd package.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
CSRC=$ROOT/sift3d_amd/csrc
OBJ=$HERE/obj
mkdir -p "$OBJ"
CXXFLAGS="-DS3D_TESTING -std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -I$HERE -I$ROOT/include -I$CSRC -Wno-attributes -Wno-unknown-pragmas"
for f in s3d_rt s3d_image s3d_gauss s3d_gauss_tab s3d_extrema s3d_keypoint s3d_dense s3d_match s3d_resample s3d_rccl; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$CSRC/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$OBJ/$f.o" ] \
     || [ "$CSRC/s3d_math.h" -nt "$OBJ/$f.o" ] || [ "$CSRC/s3d_common.h" -nt "$OBJ/$f.o" ] || [ "$CSRC/s3d_ring.h" -nt "$OBJ/$f.o" ] || [ "$ROOT/include/s3d_device.h" -nt "$OBJ/$f.o" ]; then
    g++ $CXXFLAGS -x c++ -c "$CSRC/$f.hip" -o "$OBJ/$f.o"
  fi
done
for f in s3d_host_util s3d_host_api s3d_host_match s3d_host_io s3d_host_cli s3d_host_reg s3d_host_draw s3d_host_slab s3d_host_mat; do
  gcc -DS3D_TESTING -std=gnu11 -O1 -g -fPIC -ffp-contract=off -I"$ROOT/include" -I"$CSRC/host" -pthread -c "$CSRC/host/$f.c" -o "$OBJ/$f.o"
done
g++ -shared -fPIC -Wl,-Bsymbolic -o "$HERE/libsift3d_emu.so" "$OBJ"/*.o -lm -lpthread -lz -ldl
# the in-process stand-in for librccl.so.1 (tests/emu/mock_rccl.c) that csrc/s3d_rccl.hip is driven against on the CPU
mkdir -p "$HERE/mock"
gcc -std=gnu11 -O1 -g -fPIC -shared -Wl,-soname,librccl.so.1 -o "$HERE/mock/librccl.so.1" "$HERE/mock_rccl.c" -lpthread
echo "$HERE/libsift3d_emu.so"
