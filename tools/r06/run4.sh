#!/bin/bash
# round 6, GPU call 4: full GPU suite on the staged prepare_batch / trainer graph replay, then the product-path rates
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_trainer_gpu.py tests/test_input_u8_gpu.py -x -q -m gpu -s > $out/gputest_new.log 2>&1; echo "new tests rc=$?"; tail -4 $out/gputest_new.log; grep "alternating steps, worst" $out/gputest_new.log
{ for b in 192 12; do
    timeout 600 python tools/bench_fed.py $b
    timeout 600 python tools/bench_trainer.py $b 12
    TVTS_TRAINER_GRAPH=0 timeout 600 python tools/bench_trainer.py $b 12 | sed 's/^/[TVTS_TRAINER_GRAPH=0] /'
    timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-260
  done; } 2>&1 | grep -v amdgpu.ids > $out/bench_product_path.txt
cat $out/bench_product_path.txt
timeout 1700 python -m pytest tests -x -q -m gpu > $out/gputest_full.log 2>&1; echo "full suite rc=$?"; tail -5 $out/gputest_full.log
