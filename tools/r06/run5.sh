#!/bin/bash
# round 6, GPU call 5: MFMA shape / power probe, product-path rates with longer epochs
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
timeout 300 experiments/probes/mfma_power > $out/mfma_power.txt 2>&1; cat $out/mfma_power.txt
timeout 600 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -2
{ for b in 192 12; do
    st=40; [ $b = 12 ] && st=200
    timeout 600 python tools/bench_fed.py $b
    timeout 900 python tools/bench_trainer.py $b $st
    TVTS_TRAINER_GRAPH=0 timeout 900 python tools/bench_trainer.py $b $st | sed 's/^/[TVTS_TRAINER_GRAPH=0] /'
    timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-200
    timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-graph | cut -c1-200
  done; } 2>&1 | grep -v amdgpu.ids > $out/bench_product_path.txt
cat $out/bench_product_path.txt
