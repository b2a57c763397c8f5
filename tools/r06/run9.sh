#!/bin/bash
# round 6, GPU call 9: final-tree checks -- smoke(), bench.py default + auto graph, the distributed bench tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('default', round(d['value'],1), d['config']['hip_graph'], r['frac'], r['clock_mhz_under_load'], r['frac_at_clock'], r['mfma_busy'], r['traffic'])"
python bench.py --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 pairs', round(d['value'],1), d['config']['hip_graph'])"
timeout 1200 python -m pytest tests/test_bench_dist_gpu.py tests/test_dist_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -3
