#!/bin/bash
# GPU idle gaps between consecutive kernels of eager bench steps (where the host falls behind the GPU).  usage: experiments/dbg/gaps.sh [bench args]
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_gaps
rm -rf $out; mkdir -p $out
cd /tmp
timeout 420 rocprofv3 --kernel-trace -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline "$@" > $out/run.log 2>&1
db=$(find $out -name '*_results.db' | head -1)
python - "$db" <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last 4 steps: find AdamW launches as step ends
ends = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
lo, hi = ends[-5] + 1, ends[-1] + 1
rows = rows[lo:hi]
steps = 4
busy = sum(r[2] - r[1] for r in rows)
wall = rows[-1][2] - rows[0][1]
gaps = collections.Counter(); cnt = collections.Counter()
big = []
for a, b in zip(rows[:-1], rows[1:]):
    g = b[1] - a[2]
    if g > 2000:
        key = a[0][:50] + " -> " + b[0][:50]
        gaps[key] += g; cnt[key] += 1
        big.append(g)
print(f"wall {wall/1e6/steps:.2f} ms/step, kernels busy {busy/1e6/steps:.2f} ms/step, idle {(wall-busy)/1e6/steps:.2f} ms/step; gaps > 2 us: {len(big)/steps:.0f} per step = {sum(big)/1e6/steps:.2f} ms/step")
for k, v in gaps.most_common(25):
    print(f"  {v/1e6/steps:7.3f} ms/step  n/step={cnt[k]/steps:5.1f}  {k}")
PY
rm -f $db
