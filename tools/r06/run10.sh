#!/bin/bash
# round 6, GPU call 10: the whole GPU suite and the driver-form bench on the final tree (for the record)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "import torch" > /dev/null 2>&1
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gputest_run.log 2>&1; echo "full suite rc=$?"; grep -n "passed\|failed" gpurun_out/r06/gputest_run.log
python bench.py > gpurun_out/r06/bench_default_driver_form_final.json 2>/dev/null; cut -c1-200 gpurun_out/r06/bench_default_driver_form_final.json
