#!/usr/bin/env python3
"""Timeline of the LAST detect in a rocprofv3 --kernel-trace database: every dispatch with its start offset, duration and
queue, the idle gaps on the critical path, and the totals -- where a detect's wall time goes that no kernel accounts for.
usage: python scripts/trace_timeline.py <results.db> [first-kernel-name-substring = k_absmax]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_absmax"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if first in r[0]]
if not starts:
    sys.exit("no " + first)
rows = rows[starts[-1]:]
t0 = rows[0][1]
busy_until = t0
gap_total = 0
print(f"| t0 us | dur us | queue | kernel | idle before (all queues) us |\n|---:|---:|---|---|---:|")
for r in rows:
    name = r[0].split("(")[0].replace("void ", "")
    gap = max(0, r[1] - busy_until)
    gap_total += gap
    print(f"| {(r[1]-t0)/1e3:9.1f} | {(r[2]-r[1])/1e3:8.1f} | {r[3] if qcol else ''} | `{name[:60]}` | {gap/1e3:.1f} |")
    busy_until = max(busy_until, r[2])
print(f"\nspan {(busy_until - t0)/1e3:.1f} us, GPU idle inside it {gap_total/1e3:.1f} us, {len(rows)} dispatches")
