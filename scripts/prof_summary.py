#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` result database (rocpd SQLite, the ROCm 7.2 default
output) as a per-kernel table:  python scripts/prof_summary.py gpurun_out/prof/bench_results.db > profiles/x.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                  "max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds B | scratch B |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for r in rows:
    name = r[0].split("(")[0].replace("void ", "")
    print(f"| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} |")
print(f"\ntotal kernel time {tot/1e6:.3f} ms")
