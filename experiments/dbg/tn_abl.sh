#!/bin/bash
# timing variants of the 256x256 TN kernel: rebuilds libtvts_hip.so on the GPU box with extra -D flags and times the big shapes
# usage: experiments/dbg/tn_abl.sh "-DTN_ABL=1" "-DTN_EARLY_BARRIER=1" ...
cd $GRAFT_REPO_ROOT/tvts_amd/csrc
for fl in "" "$@"; do
  touch gemm.hip
  make FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I../../include $fl" build/gemm.o ../libtvts_hip.so > /dev/null 2>&1
  echo "== flags: $fl"
  (cd ../.. && python tools/tn_ab.py 192 3 2>&1 | grep -E "qkv wgrad|fc1 wgrad|fc2 wgrad|proj wgrad" | sed -E "s/ rel [^|]*//g" | cut -c1-220)
done
touch gemm.hip; make build/gemm.o ../libtvts_hip.so > /dev/null 2>&1
