#!/bin/bash
# round 6, GPU call 3: new tests, driver-form bench (in-kernel clock sample), product-path rates (host-fed step, trainer epoch with graph replay), full GPU suite
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_trainer_gpu.py -x -q -m gpu -s -k "h14_t16_b2 or used_rows or graph_replay_in_the_trainer" > $out/gputest_new.log 2>&1; echo "new tests rc=$?"; tail -4 $out/gputest_new.log
timeout 600 python bench.py > $out/bench_default_driver_form.json 2> $out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_default_driver_form.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("achieved", "frac", "step_frac", "clock_mhz_under_load", "clock_mhz_min_max", "clock_samples", "peak_at_clock", "frac_at_clock", "mfma_busy")})
PY
{ for b in 192 12; do
    timeout 600 python tools/bench_fed.py $b
    timeout 600 python tools/bench_trainer.py $b 12
    TVTS_TRAINER_GRAPH=0 timeout 600 python tools/bench_trainer.py $b 12 | sed 's/^/[TVTS_TRAINER_GRAPH=0] /'
    timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-260
  done; } > $out/bench_product_path.txt 2>&1
cat $out/bench_product_path.txt
timeout 1700 python -m pytest tests -x -q -m gpu > $out/gputest_full.log 2>&1; echo "full suite rc=$?"; tail -5 $out/gputest_full.log
