#!/usr/bin/env python3
"""Text-tower GEMM shapes (M = NT*B*L = 24576 rows at 192 pairs): 128-tile vs 256-tile kernel choice (run once per
TVTS_NT_MIN_TILES value).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"
M = int(os.environ.get("ROWS", "24576"))


def timeit(fn, iters=50):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for n, k, f32res in ((512, 512, True), (512, 2048, True), (512, 1536, False), (512, 2048, False), (512, 512, False)):
    a = torch.randn(M, k, device=dev).bfloat16()
    b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    out = torch.empty(M, n, dtype=torch.float32 if f32res else torch.bfloat16, device=dev)
    kw = dict(bias=torch.randn(n, device=dev), residual=torch.randn(M, n, device=dev)) if f32res else {}
    ts = sorted(timeit(lambda: K.gemm_nt(a, b, out, **kw)) for _ in range(5))
    print(f"N={n} K={k} f32res={f32res}: {ts[2] * 1e3:6.1f} us {2.0 * M * n * k / ts[2] / 1e9:6.0f} TF", flush=True)
