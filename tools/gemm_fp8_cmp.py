#!/usr/bin/env python3
"""fp8 vs bf16 NT GEMM on the H/14 forward shapes (interleaved medians).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"
M = int(os.environ.get("PAIRS", "48")) * 1233


def timeit(fn, iters=8):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for n, k, act in ((3840, 1280, None), (5120, 1280, "gelu"), (1280, 5120, None), (1280, 1280, None)):
    a = torch.randn(M, k, device=dev).bfloat16()
    b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    a8, rs = K.quantize_fp8_rows(a)
    b8, sb = K.quantize_fp8(b)
    out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    pre = torch.empty(M, n, dtype=torch.bfloat16, device=dev) if act else None
    def fp8(mx):
        K.gemm_nt_fp8(a8, rs, b8, sb, out, bias=bias, act=act, preact=pre, k32=not mx)

    fns = {"bf16": lambda: K.gemm_nt(a, b, out, bias=bias, act=act, preact=pre),
           "fp8 16x16x32": lambda: fp8(False), "fp8 mx 16x16x128": lambda: fp8(True)}
    fp8(False); o0 = out.clone(); fp8(True)
    assert torch.equal(o0, out) or float((o0.float() - out.float()).abs().max()) <= 2e-2 * float(o0.float().abs().max()), "mx != 16x16x32"
    res = {name: [] for name in fns}
    for rnd in range(7):
        for name in res:
            res[name].append(timeit(fns[name]))
    line = f"N={n:5d} K={k:5d} act={act}"
    for name, ts in res.items():
        ms = sorted(ts)[3]
        line += f" | {name}: {ms * 1e3:6.1f}us {2.0 * M * n * k / ms / 1e9:5.0f}TF"
    print(line, flush=True)
