#!/usr/bin/env python3
"""LayerNorm forward / backward bandwidth at the B/16 step shape (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvts_amd import hip as K

def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

M, W = int(os.environ.get("PAIRS", "128")) * 785, int(os.environ.get("WIDTH", "768"))
dev = "cuda"
x = torch.randn(M, W, device=dev); g = torch.randn(W, device=dev); b = torch.randn(W, device=dev)
y = torch.empty(M, W, dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
t = timeit(lambda: K.layernorm_fwd(x, g, b, 1e-5, y, mean, rstd))
print(f"ln_fwd           {t:7.1f} us  {M*W*6/t/1e6:6.2f} TB/s")
dy = torch.randn(M, W, device=dev).bfloat16(); res1 = torch.randn(M, W, device=dev); res2 = torch.randn(M, W, device=dev).bfloat16()
dx = torch.empty(M, W, device=dev); dxb = torch.empty(M, W, dtype=torch.bfloat16, device=dev)
dg = torch.zeros(W, device=dev); db = torch.zeros(W, device=dev)
for name, kw, byt in (("bwd ln_2 (res1)", dict(dx=dx, dx_bf16=dxb, res1=res1), 16), ("bwd ln_1 (bf16 out)", dict(dx=None, dx_bf16=dxb), 8),
                      ("bwd ln_3 (res1+res2)", dict(dx=dx, dx_bf16=dxb, res1=res1, res2=res2), 18)):
    d = kw.pop("dx")
    t = timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, g, d, dgamma=dg, dbeta=db, **kw))
    print(f"{name:22s} {t:7.1f} us  {M*W*byt/t/1e6:6.2f} TB/s")
    t = timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, g, d, **kw))
    print(f"{name:22s} {t:7.1f} us  {M*W*byt/t/1e6:6.2f} TB/s   (no dgamma/dbeta)")

# ---- the all-bf16 forms of the hybrid residual stream (round 5): bf16 x / y, bf16 dy / x / res1 / res2 / dx, CLS rows in fp32 side arrays
S = 785
Bc = M // S
xb = x.bfloat16(); r1b = res1.bfloat16()
cls = torch.randn(Bc, W, device=dev); cls_r1 = torch.randn(Bc, W, device=dev); cls_dx = torch.empty(Bc, W, device=dev)
t = timeit(lambda: K.layernorm_fwd(xb, g, b, 1e-5, y, mean, rstd))
print(f"ln_fwd bf16 rows              {t:7.1f} us  {M*W*4/t/1e6:6.2f} TB/s")
t = timeit(lambda: K.layernorm_fwd(xb, g, b, 1e-5, y, mean, rstd, cls_x=cls, cls_period=S))
print(f"ln_fwd bf16 rows + CLS        {t:7.1f} us  {M*W*4/t/1e6:6.2f} TB/s")
for name, kw, byt in (("bwd ln_2 bf16 (res1)", dict(res1=r1b), 8), ("bwd ln_1 bf16", dict(), 6), ("bwd ln_3 bf16 (res1+res2)", dict(res1=r1b, res2=res2), 10)):
    t = timeit(lambda: K.layernorm_bwd(dy, xb, mean, rstd, g, None, dx_bf16=dxb, dgamma=dg, dbeta=db, **kw))
    print(f"{name:30s}{t:7.1f} us  {M*W*byt/t/1e6:6.2f} TB/s")
    if kw:
        t = timeit(lambda: K.layernorm_bwd(dy, xb, mean, rstd, g, None, dx_bf16=dxb, dgamma=dg, dbeta=db, cls_period=S, cls_x=cls, cls_res1=cls_r1,
                                           cls_dx=cls_dx, **kw))
        print(f"{name + ' + CLS':30s}{t:7.1f} us  {M*W*byt/t/1e6:6.2f} TB/s")
