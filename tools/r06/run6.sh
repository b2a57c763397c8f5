#!/bin/bash
# round 6, GPU call 6: trainer loop with the lazy log line (12 / 24 / 192 pairs), trainer tests
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_trainer_gpu.py tests/test_validation_gpu.py tests/test_v1_gpu.py -x -q -m gpu 2>&1 | tail -2
{ for b in 12 24 192; do
    st=200; [ $b = 192 ] && st=40
    timeout 600 python tools/bench_fed.py $b
    timeout 900 python tools/bench_trainer.py $b $st
    TVTS_TRAINER_GRAPH=1 timeout 900 python tools/bench_trainer.py $b $st | sed 's/^/[TVTS_TRAINER_GRAPH=1] /'
    timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-200
    timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-graph | cut -c1-200
  done; } 2>&1 | grep -v amdgpu.ids > $out/bench_product_path.txt
cat $out/bench_product_path.txt
