#!/usr/bin/env python3
"""L2 behaviour of the 256x256 NT kernel on the step shapes: tile order (column groups) x non-temporal A loads
(TVTS_NT_ABLATE dev knob: bit 4 = nt on A, bit 5 = nt on B, bits 8..11 = columns per group).  GPU only."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

B = int(os.environ.get("PAIRS", "192"))
M = B * 785
dev = "cuda:0"
libc = ctypes.CDLL(None)


def setenv(v):
    libc.setenv(b"TVTS_NT_ABLATE", str(v).encode(), 1)


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


variants = [int(x, 0) for x in os.environ.get("VARIANTS", "0 16 0x300 0x310 0x400 0x410 0x600 0x610 0x200 0x210").split()]
SHAPES = ((2304, 768, {}), (3072, 768, {"act": 1}), (3072, 768, {"gate": 1}), (768, 3072, {}), (768, 768, {}), (768, 2304, {}))
if os.environ.get("H14"):
    M = int(os.environ.get("PAIRS", "48")) * 1233
    SHAPES = ((3840, 1280, {}), (5120, 1280, {"act": 1}), (5120, 1280, {"gate": 1}), (1280, 5120, {}), (1280, 1280, {}), (1280, 3840, {}))
for n, k, kw in SHAPES:
    a = torch.randn(M, k, device=dev).bfloat16()
    b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    args = {}
    if kw.get("act"):
        args["act"] = "quick_gelu"; args["preact"] = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    if kw.get("gate"):
        args["gate_h"] = torch.randn(M, n, device=dev).bfloat16(); args["gate_act"] = "quick_gelu"
    ref = None
    line = f"N={n:5d} K={k:5d} {str(kw):12s}"
    out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    times = {v: [] for v in variants}
    ok = {v: True for v in variants}
    for rnd in range(int(os.environ.get("ROUNDS", "7"))):   # interleaved rounds, median: box clocks drift by several %
        for v in variants:
            setenv(v)
            times[v].append(timeit(lambda: K.gemm_nt(a, b, out, **args), iters=int(os.environ.get("ITERS", "8"))))
            if ref is None:
                ref = out.clone()
            ok[v] = ok[v] and torch.equal(ref, out)
    for v in variants:
        ms = sorted(times[v])[len(times[v]) // 2]
        line += f" | {v:#x}: {ms * 1e3:6.1f}us {2.0 * M * n * k / ms / 1e9:5.0f}TF{'' if ok[v] else ' MISMATCH'}"
    print(line, flush=True)
setenv(0)
