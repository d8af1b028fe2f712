#!/bin/bash
# GPU-box helper: kernel trace of detect per orientation mode (shipped_m<mode>) or with a variant library
# (sift3d_amd/lib/ablate/libsift3d_amd_<name>.so from scripts/build_file_variants.py, mode ORI_MODE)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/orient_exp.txt
: > $out
for v in ${VARIANTS_O:-shipped_m0 shipped_m1 shipped_m2}; do
  lib=$R/sift3d_amd/lib/ablate/libsift3d_amd_$v.so
  case "$v" in shipped*) lib=$R/sift3d_amd/lib/libsift3d_amd.so;; esac
  m=${ORI_MODE:-2}
  case "$v" in shipped_m*) m=${v#shipped_m};; esac
  echo "== $v" >> $out
  ( cd /tmp && export TMPDIR=/tmp && S3D_ORI_MODE=$m SIFT3D_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_oexp_$v" -o d -- python "$R/scripts/detect_ab.py" > "$R/gpurun_out/prof_oexp_$v.log" 2>&1 )
  grep "detect min" gpurun_out/prof_oexp_$v.log >> $out
  python - "$R/gpurun_out/prof_oexp_$v" >> $out <<'PY'
import sqlite3, sys, re, glob, collections
db = glob.glob(sys.argv[1] + "/*.db")[0]
c = sqlite3.connect(db)
d = collections.defaultdict(list)
for name, s, e in c.execute("select name,start,end from kernels"):
    d[re.sub(r"\(.*", "", name)[:36]].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "orient" in k: print(f"  {k:38s} n={len(v):3d} avg {sum(v)/len(v):8.1f} min {min(v):8.1f} us")
PY
done
cat $out
