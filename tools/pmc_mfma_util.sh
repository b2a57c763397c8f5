#!/bin/bash
# MFMA utilisation, wait / issue-stall shares and the clock under load of the kernels of ONE bench step (BASELINE north_star:
# "evidenced by rocprof HBM GB/s and MFMA utilisation against gfx950 peak"; SURVEY 8d: "confirm the peak on the box").
# Separate rocprofv3 --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, "rocprofv3 PMC slots": 8 SQ slots, 2 GRBM):
#   pass sq   : SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
#               SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS
#   pass grbm : GRBM_GUI_ACTIVE GRBM_COUNT                       (clock = GUI_ACTIVE cycles / the dispatch's traced duration)
# usage (on the GPU box): tools/pmc_mfma_util.sh [bench.py flags]   -> gpurun_out/pmc_mfma_util.{json,txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch" > /dev/null 2>&1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS"
for p in sq grbm; do
  rm -rf gpurun_out/pmcu_$p
  if [ $p = sq ]; then C="$SQ"; else C="GRBM_GUI_ACTIVE GRBM_COUNT"; fi
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmcu_$p -o p -- \
    python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --text-side off "$@" > gpurun_out/pmcu_$p.log 2>&1
  echo "pass $p rc=$?"
done
TVTS_BENCH_ORDER=gpurun_out/gemm_order_util.json python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --text-side off "$@" > /dev/null 2> gpurun_out/pmcu_order.err
python tools/pmc_mfma_join.py gpurun_out/gemm_order_util.json gpurun_out "$@" | tee gpurun_out/pmc_mfma_util.txt
