#!/bin/bash
# usage: experiments/dbg/kasm.sh <asm file> <mangled kernel name> -> prints memory ops / waits / barriers of that kernel in order
awk "/^$2:/,/s_endpgm/" $1 > /tmp/kasm_k.s
wc -l /tmp/kasm_k.s
grep -n "s_waitcnt vm\|s_waitcnt lgkmcnt(0) vm\|s_barrier\|global_load\|global_store\|scratch\|;;#ASM\|s_cbranch\|^.LBB" /tmp/kasm_k.s | awk '{print $1,$2,$3,$4}' | cut -c1-64
