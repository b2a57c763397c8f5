#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "import torch" > /dev/null 2>&1
AB_VARIANTS="pasm+pin,b16 pasm,b16 side" timeout 900 python experiments/gemm_ab.py 192 7 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/gemm_ab_b16_side.txt; cat gpurun_out/r06/gemm_ab_b16_side.txt
