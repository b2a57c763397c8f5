#!/usr/bin/env python3
"""Timing ablations of the fused divided-attention backward at the B/16 step shape (dev tool, GPU): the kernel with phase A
(dQ), phase B (dK / dV) and the global loads switched off in turn (results are wrong by construction, tvts_attn_bwd opts bits 4..6).
PAIRS=192 python tools/attn_ablate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

B = int(os.environ.get("PAIRS", "192"))
dev = "cuda:0"


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for mode, T, n in (("space", 8, 98), ("time", 8, 98)):
    heads, S = 12, 785
    W, M = heads * 64, B * 785
    qkv = torch.randn(M, 3 * W, device=dev).bfloat16()
    dO = torch.randn(M, W, device=dev).bfloat16()
    out = torch.empty(M, W, dtype=torch.bfloat16, device=dev)
    lse, delta = torch.empty(M, heads, device=dev), torch.empty(M, heads, device=dev)
    dqkv = torch.empty(M, 3 * W, dtype=torch.bfloat16, device=dev)
    acc = torch.zeros(B, heads, max(T, -(-n // 28), 1), 3, 64, device=dev)
    ws = torch.empty(B * heads * max(T, -(-n // 28)) * 66, device=dev)
    kw = dict(B=B, heads=heads, S=S, T=T, n=n)
    K.attn_fwd_divided(mode, qkv, out, lse, ws, **kw)
    gb = (M * 3 * W * 2 * 2 + M * W * 2 * 2) / 1e9
    for ab, label in ((0, "full kernel"), (1, "no phase A"), (2, "no phase B"), (3, "no phase A, B (loads + LDS staging + launch)"), (4, "no global loads"),
                      (7, "nothing but LDS staging of zeros")):
        with K.options(attn_ablate=ab):
            t = timeit(lambda: K.attn_bwd(mode, qkv, dO, out, lse, delta, dqkv, cls_acc=acc, **kw))
        print(f"{mode:6s} bwd ablate={ab} {label:48s} {t:7.1f} us   ({gb / t * 1e3:5.2f} TB/s on the full kernel's {gb:.2f} GB)")
    tf = timeit(lambda: K.attn_fwd_divided(mode, qkv, out, lse, ws, **kw))
    gf = (M * 3 * W * 2 + M * W * 2) / 1e9
    print(f"{mode:6s} fwd {tf:7.1f} us ({gf / tf * 1e3:5.2f} TB/s on {gf:.2f} GB)")
