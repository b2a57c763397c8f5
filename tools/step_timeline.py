#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> the launches of ONE steady step in start order: duration, gap to the previous kernel's end, and the
sum of both per kernel family.  Answers "kernel time or gaps?" for the small per-GPU batches.
usage: step_timeline.py <results.db> [n_steps_traced] [which_step] > timeline.txt
The step boundary is the adamw_kernel launch (one per step)."""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
cuts = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
if len(cuts) < 3:
    sys.exit(f"only {len(cuts)} adamw launches in the trace")
lo, hi = cuts[which - 1] + 1, cuts[which] + 1
step = rows[lo:hi]
t0 = step[0][1]
wall = step[-1][2] - t0
busy = sum(r[2] - r[1] for r in step)
print(f"# step of {len(step)} launches: wall {wall / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, gaps {(wall - busy) / 1e6:.3f} ms")
fam = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = t0
for name, s, e in step:
    gap = s - prev_end
    prev_end = max(prev_end, e)
    short = name.split("(")[0][:70]
    f = fam[short]
    f[0] += 1
    f[1] += e - s
    f[2] += max(gap, 0)
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short}")
print("# per kernel: launches, kernel ms, gap-before ms")
for k, (n, d, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"# {n:5d} {d / 1e6:8.3f} {g / 1e6:8.3f}  {k}")
