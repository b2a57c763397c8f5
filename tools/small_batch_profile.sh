#!/bin/bash
# Small per-GPU batches (the reference's own 12 / 24 pairs, B/32 at 24): bench lines + a graph-replay kernel timeline of one step.
# usage (GPU box, repo root): tools/small_batch_profile.sh <tag>
tag=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
{ python bench.py --batch 2 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  python bench.py --batch 12 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  python bench.py --batch 24 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  python bench.py --arch B_32 --batch 24 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline; } 2>$out/bench_small.err | grep '^{' > $out/bench_reference_batches.jsonl
for b in 12 24; do
  d=$out/trace_b$b
  mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $d -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --text-side off > $d/run.log 2>&1)
  db=$(find $d -name '*_results.db' | head -1)
  python tools/prof_summary.py $db $d/kernel_stats.csv 6 > $out/kernel_summary_b$b.txt 2>&1
  python tools/step_timeline.py $db > $out/timeline_b$b.txt 2>&1
  rm -rf $d
done
cut -c1-220 $out/bench_reference_batches.jsonl
head -5 $out/timeline_b12.txt; grep '^#' $out/timeline_b12.txt | head -40
