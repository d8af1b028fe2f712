#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
( cd /tmp && export TMPDIR=/tmp && REPS=4 timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/tl" -o t -- python "$R/scripts/detect_one.py" > "$R/gpurun_out/tl.log" 2>&1 )
f=$(find gpurun_out/tl -name "*.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print([r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")][:60])
print([r[1] for r in db.execute("pragma table_info(kernels)")])
PY
python scripts/trace_timeline.py $f > gpurun_out/detect_timeline.md
tail -n 1 gpurun_out/tl.log; tail -n 3 gpurun_out/detect_timeline.md
rm -rf gpurun_out/tl
