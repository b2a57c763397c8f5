#!/bin/bash
# The round's measurement set on one MI355X (run through gpurun from the repo root):  tools/round_profile.sh r02
#   1. rocprofv3 --kernel-trace of the default bench step (eager launches)  -> gpurun_out/<tag>/kernel_summary_default_b192.txt
#   2. tools/pmc_traffic.sh (FETCH_SIZE / WRITE_SIZE passes) + the per-shape join -> pmc_step_traffic_*.json, pmc_gemm_traffic_by_shape.txt
#   3. bench lines: default, the reference's per-GPU batch (12 pairs), WebVid-style NT = 1, B/32 (configs[1]), H/14 16 frames
#      bf16 / fp8 forward / + input gradients / + weight gradients (+ its kernel trace, the fp8 GEMM and attention micro-benchmarks), v1 -> bench_*.json(l)
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -c "import torch" > /dev/null 2>&1
# the counter passes first: the bench line quotes them (roofline.traffic, roofline.mfma_busy) and refuses files measured on other GEMM sources
tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_step_traffic.json $out/pmc_step_traffic_B_16_t8_b192.json
cp gpurun_out/pmc_step_traffic.json profiles/${tag}_pmc_step_traffic_B_16_t8_b192.json
tools/pmc_mfma_util.sh > $out/pmc_mfma_util.log 2>&1
cp gpurun_out/pmc_mfma_util.txt $out/pmc_mfma_util_B_16_t8_b192.txt
cp gpurun_out/pmc_mfma_util.json $out/pmc_mfma_util_B_16_t8_b192.json
cp gpurun_out/pmc_mfma_util.json profiles/${tag}_pmc_mfma_util_B_16_t8_b192.json
python bench.py > $out/bench_default_driver_form.json 2> $out/bench_default.err                     # exactly what the driver runs
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_default_b192.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --graph 2>/dev/null | grep '^{' > $out/bench_default_b192_graph_replay.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dense-sort-head 2>/dev/null | grep '^{' > $out/bench_dense_sort_head.json
# (the text tower in line: in the timed steps it runs on its own stream beside the ViT, and a kernel's traced duration would then
#  include the time it shared the chip with the other stream's kernels -- same kernels, same bits either way)
tools/profile_step.sh ${tag}_default --text-side off > $out/profile_step.log 2>&1
cp gpurun_out/prof_${tag}_default/summary.txt $out/kernel_summary_default_b192.txt
cp gpurun_out/prof_${tag}_default/kernel_stats.csv $out/kernel_stats_default_b192.csv
[ -x experiments/probes/mfma_power ] || make -C experiments probes > /dev/null 2>&1
experiments/probes/mfma_power > $out/mfma_power_probe.txt 2>&1
TVTS_BENCH_ORDER=gpurun_out/gemm_order.json python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $out/order.err
python tools/pmc_join.py gpurun_out/gemm_order.json gpurun_out > $out/pmc_gemm_traffic_by_shape.txt 2>> $out/order.err
{ python bench.py --batch 2 --steps 20 --warmup 5 --no-cpu-baseline
  python bench.py --batch 12 --steps 20 --warmup 5 --no-cpu-baseline
  python bench.py --batch 24 --steps 20 --warmup 5 --no-cpu-baseline
  python bench.py --n-trans 1 --steps 20 --warmup 5 --no-cpu-baseline; } 2>/dev/null | grep '^{' > $out/bench_reference_batches.jsonl
{ python bench.py --arch B_32 --batch 384 --steps 20 --warmup 5 --no-cpu-baseline
  python bench.py --arch B_32 --batch 24 --steps 20 --warmup 5 --no-cpu-baseline; } 2>/dev/null | grep '^{' > $out/bench_b32_t8.jsonl
{ python bench.py --arch H_14 --frames 16 --batch 48 --steps 8 --warmup 3 --no-cpu-baseline
  python bench.py --arch H_14 --frames 16 --batch 48 --steps 8 --warmup 3 --no-cpu-baseline --fp8
  python bench.py --arch H_14 --frames 16 --batch 48 --steps 8 --warmup 3 --no-cpu-baseline --fp8-dgrad
  python bench.py --arch H_14 --frames 16 --batch 48 --steps 8 --warmup 3 --no-cpu-baseline --fp8-wgrad; } 2>/dev/null | grep '^{' > $out/bench_h14_t16_b48.jsonl
tools/profile_step.sh ${tag}_h14fp8 --arch H_14 --frames 16 --batch 48 --fp8-wgrad --text-side off > $out/profile_h14fp8.log 2>&1
cp gpurun_out/prof_${tag}_h14fp8/summary.txt $out/kernel_summary_h14_b48_fp8_wgrad.txt
python tools/tn_fp8_bench.py > $out/tn_fp8_bench.txt 2>&1
PAIRS=48 python tools/gemm_fp8_cmp.py > $out/gemm_fp8_cmp.txt 2>&1
PAIRS=192 python tools/attn_bench.py > $out/attn_bench.txt 2>&1
{ python bench.py --arch v1 --frames 4 --batch 256 --steps 20 --warmup 5 --cpu-pairs 8
  python bench.py --arch v1 --frames 16 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline; } 2>/dev/null | grep '^{' > $out/bench_v1.jsonl
PAIRS=192 python tools/attn_ablate.py > $out/attn_ablate.txt 2>&1
tools/profile_step.sh ${tag}_v1 --arch v1 --frames 4 --batch 256 > $out/profile_v1.log 2>&1
cp gpurun_out/prof_${tag}_v1/summary.txt $out/kernel_summary_v1_t4_b256.txt
ls -la $out
head -12 $out/kernel_summary_default_b192.txt
cut -c1-160 $out/bench_default_b192.json | tail -1
