#!/bin/bash
# The 1 / 2 / 4 / 8-GPU bench lines of one node, for the day a multi-GPU box exists (none was available to rounds 1-4):
#   tools/scale_run.sh [out.jsonl] [bench args...]
# For every N: the default exchange path (torch.distributed over RCCL, eager launches), then the native transport
# (TVTS_COMM=native: libtvts_comm.so, RCCL on the library's side stream) eagerly and with the whole multi-rank step captured into a
# hipGraph (TVTS_BENCH_GRAPH_DDP=1).  Every line carries config.exchange.ranks_seen / devices_seen (an all-reduce in which each
# rank adds 1 and its device index) and, for the native transport, the result of its known-answer self-test -- proof that N ranks
# on N devices took part.  Efficiency is for the reader to compute from the per-N values.
out=${1:-gpurun_out/scale_run.jsonl}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
: > $out
for n in 1 2 4 8; do
  [ $n -gt $ngpu ] && { echo "# $n GPUs: only $ngpu visible, skipped" >> $out; continue; }
  for mode in torch native native_graph; do
    [ $n -eq 1 ] && [ $mode != torch ] && continue
    env=""
    [ $mode = native ] && env="TVTS_COMM=native"
    [ $mode = native_graph ] && env="TVTS_COMM=native TVTS_BENCH_GRAPH_DDP=1"
    port=$((29500 + n * 10 + ${#mode}))
    if [ $n -eq 1 ]; then cmd="python bench.py --gpus 1"; else
      cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n"; fi
    gflag=""; [ $mode = native_graph ] && gflag="--graph"   # (eager launches are bench.py's default since round 6)
    echo "# N=$n mode=$mode" >> $out
    env $env timeout 900 $cmd --steps 10 --warmup 3 --no-cpu-baseline $gflag "$@" 2>/dev/null | grep '^{' >> $out || echo "# failed or timed out" >> $out
  done
done
cat $out | cut -c1-200
