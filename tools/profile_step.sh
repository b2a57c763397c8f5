#!/bin/bash
# Kernel trace of a few bench steps (eager launches so every kernel shows individually), summarised per kernel.
# usage (on the GPU box): tools/profile_step.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
timeout 420 rocprofv3 --kernel-trace -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline "$@" > $out/run.log 2>&1
db=$(find $out -name '*_results.db' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db $out/kernel_stats.csv 6 > $out/summary.txt 2>&1
tail -3 $out/run.log
cat $out/summary.txt
rm -f $db
