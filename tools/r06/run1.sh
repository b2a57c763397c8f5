#!/bin/bash
# round 6, GPU call 1: MFMA-utilisation / clock counters on the bench step, H/14 16-frame sizing runs, driver-form bench
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
timeout 1200 tools/pmc_mfma_util.sh > $out/pmc_mfma_util.log 2>&1
cp gpurun_out/pmc_mfma_util.txt $out/pmc_mfma_util_B_16_t8_b192.txt
cp gpurun_out/pmc_mfma_util.json $out/pmc_mfma_util_B_16_t8_b192.json
tail -5 $out/pmc_mfma_util.log
for b in 64 80 96; do
  timeout 600 python bench.py --arch H_14 --frames 16 --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/h14_b$b.err | grep '^{' >> $out/bench_h14_max_batch.jsonl || { echo "H/14 batch $b failed"; tail -3 $out/h14_b$b.err; break; }
  tail -1 $out/bench_h14_max_batch.jsonl | cut -c1-200
done
timeout 600 python bench.py > $out/bench_default_driver_form_base.json 2> $out/bench_default_base.err
cut -c1-300 $out/bench_default_driver_form_base.json
