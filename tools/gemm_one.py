#!/usr/bin/env python3
"""Run one NT (or TN) GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py nt|tn M N K [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

kind, M, N, Kd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = "cuda:0"
if kind == "nt":
    a = torch.randn(M, Kd, device=dev).bfloat16()
    b = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(iters):
        K.gemm_nt(a, b, out)
else:
    p = torch.randn(M, N, device=dev).bfloat16()
    q = torch.randn(M, Kd, device=dev).bfloat16()
    out = torch.zeros(N, Kd, device=dev)
    for _ in range(iters):
        K.gemm_tn(p, q, out, accumulate=True)
torch.cuda.synchronize()
