#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` result database (rocpd SQLite, the ROCm 7.2 default
output) as a per-kernel table:  python scripts/prof_summary.py gpurun_out/prof/bench_results.db > profiles/x.md
Since round 6 every kernel has two more columns: how many of its dispatches ran ALONE (no other dispatch of the process overlaps
them in time: the detect runs octaves >= 1 and the extrema pass on streams of their own beside octave 0's kernels) and their
average duration -- the figure to hold against HIP-event timings of a kernel launched by itself (bench.py's roofline leg)."""
import bisect
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
disp = db.execute("select name, start, end, vgpr_count, lds_size, scratch_size from kernels order by start").fetchall()
starts = [d[1] for d in disp]
# running maximum of the end times of the dispatches before i, and the earliest start after i: overlap tests in O(log n)
max_end_before, m = [], 0
for d in disp:
    max_end_before.append(m)
    m = max(m, d[2])
agg = {}
for i, (name, s, e, vg, lds, scr) in enumerate(disp):
    j = bisect.bisect_right(starts, s, lo=i + 1)          # dispatches that start at the same instant
    nxt = starts[i + 1] if i + 1 < len(disp) else None
    alone = max_end_before[i] <= s and (nxt is None or nxt >= e) and j == i + 1
    a = agg.setdefault(name, [0, 0, 10**18, 0, 0, 0, 0, 0, 0])
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    a[4] = max(a[4], vg or 0); a[5] = max(a[5], lds or 0); a[6] = max(a[6], scr or 0)
    if alone:
        a[7] += 1; a[8] += e - s
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(a[1] for _, a in rows) or 1
print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds B | scratch B | alone: calls | alone: avg us |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for name, a in rows:
    n = name.split("(")[0].replace("void ", "")
    al = f"{a[8] / a[7] / 1e3:.2f}" if a[7] else "--"
    print(f"| `{n}` | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/tot:.1f} | {a[4]} | {a[5]} | {a[6]} | {a[7]} | {al} |")
print(f"\ntotal kernel time {tot/1e6:.3f} ms")
