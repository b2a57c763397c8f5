#!/bin/bash
# A/B of compile-time variants of the GEMM kernels on ONE box: rebuilds libtvts_hip.so with each extra -D flag set and runs a command.
# usage: experiments/dbg/ab_flags.sh '<command>' "" "-DTVTS_NT_SD=163840" ...      (the first, empty, set is the production build)
cmd="$1"; shift
cd $GRAFT_REPO_ROOT/tvts_amd/csrc
for fl in "$@"; do
  touch gemm.hip
  make FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I../../include $fl" build/gemm.o ../libtvts_hip.so > /dev/null 2>&1
  echo "== flags: $fl"
  (cd ../.. && bash -c "$cmd")
done
touch gemm.hip; make build/gemm.o ../libtvts_hip.so > /dev/null 2>&1
