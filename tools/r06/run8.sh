#!/bin/bash
# round 6, GPU call 8: eager against replayed graph where the step is launch-bound (2 pairs of B/16, 2 pairs of H/14 at 16 frames)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "import torch" > /dev/null 2>&1
{ for rep in 1 2; do
  for fl in "--no-graph" "--graph"; do
    python bench.py --batch 2 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline $fl | cut -c1-190
    python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline $fl | cut -c1-190
    python bench.py --batch 6 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline $fl | cut -c1-190
    python bench.py --arch H_14 --frames 16 --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $fl | cut -c1-190
  done; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/bench_launch_bound_eager_vs_graph.txt
cat gpurun_out/r06/bench_launch_bound_eager_vs_graph.txt
