#!/usr/bin/env python3
"""Micro-benchmark of the attention entry points at the B/16 step shapes (dev tool, GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

B = int(os.environ.get("PAIRS", "64"))


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


def run(mode, Bn, heads, S, T=0, n=0, causal=False):
    dev = "cuda:0"
    W = heads * 64
    M = Bn * S
    qkv = torch.randn(M, 3 * W, device=dev).bfloat16()
    dO = torch.randn(M, W, device=dev).bfloat16()
    out = torch.empty(M, W, dtype=torch.bfloat16, device=dev)
    lse, delta = torch.empty(M, heads, device=dev), torch.empty(M, heads, device=dev)
    dqkv = torch.empty(M, 3 * W, dtype=torch.bfloat16, device=dev)
    acc = torch.zeros(Bn, heads, max(T, -(-n // 28), 1), 3, 64, device=dev)
    kw = dict(B=Bn, heads=heads, S=S, T=T, n=n)
    if mode == "full":
        kw["causal"] = causal
    f = timeit(lambda: K.attn_fwd(mode, qkv, out, lse, **kw))
    K.attn_delta(dO, out, delta, rows=M, heads=heads)
    dq = timeit(lambda: K.attn_bwd_dq(mode, qkv, dO, lse, delta, dqkv, **kw))
    dkv = 0.0
    if mode != "cls":
        extra = dict(cls_acc=acc) if mode in ("space", "time") else {}
        dkv = timeit(lambda: K.attn_bwd_dkv(mode, qkv, dO, lse, delta, dqkv, **kw, **extra))
    site = ""
    if mode in ("space", "time"):
        ws = torch.empty(Bn * heads * max(T, -(-n // 28)) * 66, device=dev)
        tf = []
        for fused in (True, False):
            tf.append(timeit(lambda: K.attn_fwd_divided(mode, qkv, out, lse, ws, B=Bn, heads=heads, S=S, T=T, n=n, fused=fused)))
        site = f"  site-fwd fused {tf[0]:7.1f} us / split {tf[1]:7.1f} us\n"
    if mode == "full" and S <= 32:  # short sequences: wave-per-group kernels vs the streaming ones
        fs = timeit(lambda: K.attn_fwd(mode, qkv, out, lse, fused=False, **kw))
        site = f"  fwd fused {f:7.1f} us / streaming {fs:7.1f} us\n"
    if mode != "cls":
        extra = dict(cls_acc=acc) if mode in ("space", "time") else {}
        K.attn_fwd(mode, qkv, out, lse, **kw)
        ts = []
        for fused in (True, False):
            ts.append(timeit(lambda: K.attn_bwd(mode, qkv, dO, out, lse, delta, dqkv, fused=fused, **kw, **extra)))
        site += f"  site-bwd fused {ts[0]:7.1f} us / split {ts[1]:7.1f} us\n"
    gb = M * 3 * W * 2 / 1e9
    print(site, end="")
    print(f"{mode:6s} B={Bn} h={heads} S={S}: fwd {f:7.1f} us  dq {dq:7.1f} us  dkv {dkv:7.1f} us   (qkv {gb * 1e3:.0f} MB)")


run("time", B, 12, 785, 8, 98)
run("space", B, 12, 785, 8, 98)
run("cls", B, 12, 785, 8, 98)
run("full", B * 4, 8, 32, causal=True)
run("full", B, 8, 789)
