#!/usr/bin/env python3
"""Per-kernel sums of the counters in one or more rocprofv3 --pmc result databases (rocpd sqlite).
usage: pmc_summary.py results.db [...]   -> markdown table on stdout."""
import collections
import sqlite3
import sys

res = collections.defaultdict(dict)
for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    info = [t for t in tabs if "info_pmc" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct e.event_id) from {pmc} e join {info} i on e.pmc_id=i.id "
         f"join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by s.kernel_name, i.name")
    for k, n, v, c in cur.execute(q):
        res[k][n] = v
        res[k]["calls"] = c
names = sorted({n for d in res.values() for n in d if n != "calls"})
print("| kernel | calls | " + " | ".join(names) + " |")
print("|---|---:|" + "---:|" * len(names))
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_INSTS_LDS", 0)))[:14]:
    print(f"| `{k[:40]}` | {d['calls']} | " + " | ".join("%.3g" % d.get(n, 0) for n in names) + " |")
