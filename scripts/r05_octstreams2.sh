#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=$R/sift3d_amd/lib/libsift3d_amd_testing.so
for rep in 1 2 3; do
  SIFT3D_AMD_LIB=$T S3D_NO_OCTAVE_STREAMS=1 python scripts/detect_ab.py 2>&1 | tail -1 | sed 's/^/one stream              : /'
  SIFT3D_AMD_LIB=$T S3D_OCTAVE_STREAMS_PLAIN=1 python scripts/detect_ab.py 2>&1 | tail -1 | sed 's/^/octave streams, plain   : /'
  SIFT3D_AMD_LIB=$T python scripts/detect_ab.py 2>&1 | tail -1 | sed 's/^/octave streams, priority: /'
done | tee gpurun_out/octstreams2.txt
( cd /tmp && export TMPDIR=/tmp && REPS=4 timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/tl" -o t -- python "$R/scripts/detect_one.py" > "$R/gpurun_out/tl.log" 2>&1 )
f=$(find gpurun_out/tl -name "*.db" | head -1)
python scripts/trace_timeline.py $f > gpurun_out/detect_timeline.md
tail -n 1 gpurun_out/tl.log; tail -n 2 gpurun_out/detect_timeline.md
rm -rf gpurun_out/tl
