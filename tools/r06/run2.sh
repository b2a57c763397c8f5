#!/bin/bash
# round 6, GPU call 2: the new tests first, then the whole GPU suite, then the driver-form bench (live clock probe)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "h14_t16_b2 or used_rows or webvid or small_arch_forward" > $out/gputest_new.log 2>&1; echo "new tests rc=$?"; tail -8 $out/gputest_new.log
timeout 600 python bench.py > $out/bench_default_driver_form.json 2> $out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_default_driver_form.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("achieved", "frac", "step_frac", "clock_mhz_under_load", "clock_mhz_min_max", "clock_probes", "peak_at_clock", "frac_at_clock", "mfma_busy", "sheet_clock_mhz", "cu_count")})
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gputest_full.log 2>&1; echo "full suite rc=$?"; tail -5 $out/gputest_full.log
