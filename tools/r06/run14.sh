#!/bin/bash
# round 6, GPU call 14: bf16-first patch in the production dispatch -- parity tests, then the step A/B (alternating runs, one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_bench_path_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "nt256p or gemm_nt or two_threads or b16_step or bench_dispatch" 2>&1 | tail -3
{ for rep in 1 2 3; do
    echo "# bf16-first patch (default)"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-170
    echo "# fp32 patch (TVTS_NT_F32_PATCH=1)"; TVTS_NT_F32_PATCH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-170
  done
  for b in 12 24; do
    echo "# $b pairs: default / fp32 patch"; python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-170
    TVTS_NT_F32_PATCH=1 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-170
  done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/bench_b16_patch_ab.txt
cat gpurun_out/r06/bench_b16_patch_ab.txt
