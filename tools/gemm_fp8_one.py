#!/usr/bin/env python3
"""Run one e4m3 NT GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_fp8_one.py M N K [iters] [mx=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

M, N, Kd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
K.set_default(fp8_k32=not (bool(int(sys.argv[5])) if len(sys.argv) > 5 else True))
dev = "cuda:0"
a = torch.randn(M, Kd, device=dev).bfloat16()
b = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
a8, rs = K.quantize_fp8_rows(a)
b8, sb = K.quantize_fp8(b)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(iters):
    K.gemm_nt_fp8(a8, rs, b8, sb, out)
torch.cuda.synchronize()
