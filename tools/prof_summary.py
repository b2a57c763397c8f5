#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> per-kernel stats CSV, the same columns as `--stats` kernel_stats.csv.
usage: prof_summary.py <results.db> <out.csv> [steps]"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                 "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for r in rows:
        f.write(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100 * r[2] / tot:.2f},{r[4]},{r[5]}\n')
print(f"total {tot / 1e6:.2f} ms over {steps:g} steps = {tot / 1e6 / steps:.2f} ms/step")
for r in rows[:32]:
    print(f"{r[2] / 1e6 / steps:8.2f} ms/step {100 * r[2] / tot:5.1f}%  n/step={r[1] / steps:6.1f} avg={r[3] / 1e3:8.1f}us  {r[0][:90]}")
