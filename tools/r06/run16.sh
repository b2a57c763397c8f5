#!/bin/bash
# round 6, GPU call 16: re-stamp the counter files on the final GEMM sources (the header gained the not-adopted side-input experiment:
# same production kernels, new source hash), then the driver-form bench that quotes them
cd $GRAFT_REPO_ROOT
tag=r06; out=gpurun_out/$tag; mkdir -p $out
python -c "import torch" > /dev/null 2>&1
tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_step_traffic.json $out/pmc_step_traffic_B_16_t8_b192.json
cp gpurun_out/pmc_step_traffic.json profiles/${tag}_pmc_step_traffic_B_16_t8_b192.json
tools/pmc_mfma_util.sh > $out/pmc_mfma_util.log 2>&1
cp gpurun_out/pmc_mfma_util.txt $out/pmc_mfma_util_B_16_t8_b192.txt
cp gpurun_out/pmc_mfma_util.json $out/pmc_mfma_util_B_16_t8_b192.json
cp gpurun_out/pmc_mfma_util.json profiles/${tag}_pmc_mfma_util_B_16_t8_b192.json
python bench.py > $out/bench_default_driver_form.json 2> $out/bench_default.err
TVTS_BENCH_ORDER=gpurun_out/gemm_order.json python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $out/order.err
python tools/pmc_join.py gpurun_out/gemm_order.json gpurun_out > $out/pmc_gemm_traffic_by_shape.txt 2>> $out/order.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_default_driver_form.json").read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("achieved", "frac", "step_frac", "clock_mhz_under_load", "frac_at_clock", "mfma_busy", "traffic", "traffic_source")})
PY
tail -1 $out/pmc_mfma_util_B_16_t8_b192.txt; tail -1 $out/pmc_gemm_traffic_by_shape.txt
