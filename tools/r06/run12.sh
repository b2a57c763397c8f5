#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "import torch" > /dev/null 2>&1
TRACE_WIDE=1 timeout 600 python experiments/gemm_trace.py 192 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/gemm_trace_wide_forms.txt; cat gpurun_out/r06/gemm_trace_wide_forms.txt
