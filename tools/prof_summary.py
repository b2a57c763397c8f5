#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> per-kernel stats CSV, the same columns as `--stats` kernel_stats.csv.
usage: prof_summary.py <results.db> <out.csv> [steps]

Only the STEADY steps are counted: the launches between the first and the last adamw_kernel of the trace (one per optimizer step), so
that the model set-up of the traced process -- ~700 parameter copies (__amd_rocclr_copyBuffer), initial casts and transposes --
does not appear as "per step" work (round 4's summaries divided them by the step count: 113 copies "per step" that no step makes).
Without three adamw launches the whole trace is summarised over the given step count."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
cuts = [r[0] for r in c.execute("select end from kernels where name like 'adamw_kernel%' order by start").fetchall()]
where, note = "", "whole trace"
if len(cuts) >= 3:
    where = f" where start > {cuts[0]} and end <= {cuts[-1]}"
    steps = float(len(cuts) - 1)
    note = f"the {int(steps)} steps between the first and the last adamw_kernel launch (set-up and the first step excluded)"
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels" + where +
                 " group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for r in rows:
        f.write(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100 * r[2] / tot:.2f},{r[4]},{r[5]}\n')
print(f"total {tot / 1e6:.2f} ms over {steps:g} steps = {tot / 1e6 / steps:.2f} ms/step   ({note})")
for r in rows[:40]:
    print(f"{r[2] / 1e6 / steps:8.2f} ms/step {100 * r[2] / tot:5.1f}%  n/step={r[1] / steps:6.1f} avg={r[3] / 1e3:8.1f}us  {r[0][:90]}")
