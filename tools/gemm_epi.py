#!/usr/bin/env python3
"""Micro-benchmark of the NT GEMM epilogue variants on the B/16 step shapes (dev tool, GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

B = int(os.environ.get("PAIRS", "128"))
M = B * 785
dev = "cuda:0"


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(name, n, k, **kw):
    a = torch.randn(M, k, device=dev).bfloat16()
    b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    f32 = kw.pop("f32", False)
    out = torch.empty(M, n, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    args = dict(bias=bias)
    if kw.get("res"):
        args["residual"] = torch.randn(M, n, device=dev)
    if kw.get("act"):
        args["act"] = "quick_gelu"; args["preact"] = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    if kw.get("gate"):
        args["gate_h"] = torch.randn(M, n, device=dev).bfloat16(); args["gate_act"] = "quick_gelu"; args.pop("bias")
    ms = timeit(lambda: K.gemm_nt(a, b, out, **args))
    print(f"{name:22s} N={n:5d} K={k:5d}: {ms * 1e3:8.1f} us  {2.0 * M * n * k / ms / 1e9:7.1f} TF")


run("plain bf16", 768, 768)
run("f32 + residual", 768, 768, f32=True, res=True)
run("f32 only", 768, 768, f32=True)
run("bf16 + residual", 768, 768, res=True)
run("plain bf16", 3072, 768)
run("act + preact", 3072, 768, act=True)
run("gate", 3072, 768, gate=True)
run("plain bf16", 768, 3072)
run("f32 + residual", 768, 3072, f32=True, res=True)
run("plain bf16", 2304, 768)
