#!/bin/bash
# GPU box: where the waves of k_nn_gemm spend their cycles (SQ counters, one --pmc pass, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$R/gpurun_out/pmc_match" -o pmc -- python "$R/scripts/match_ab.py" > "$R/gpurun_out/pmc_match.log" 2>&1
tail -n 2 "$R/gpurun_out/pmc_match.log"
python - <<PY
import glob, sqlite3, collections
for db in glob.glob("$R/gpurun_out/pmc_match/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]; info = [t for t in tabs if "info_pmc" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct e.event_id) from {pmc} e join {info} i on e.pmc_id=i.id "
         f"join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by s.kernel_name, i.name")
    res = collections.defaultdict(dict)
    for k, c, v, n in cur.execute(q):
        res[k[:40]][c] = (v / max(n, 1))
    for k, d in res.items():
        if "gemm" in k or "verify" in k or "col_cand" in k:
            print(k, {c: f"{v:.3e}" for c, v in d.items()})
PY
