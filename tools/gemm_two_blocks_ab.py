"""One persistent 256 x 256 block per CU against TWO independent blocks per CU (the 128 x 128 kernel: 64 KiB of LDS and 134 registers per
block) on the step's epilogue forms at the bench's row count (M = 192 x 785), three rotating buffer sets, medians -- the question of
the round-4 review's item 2b: does a partner block's K loop hide a block's epilogue?  The 128 x 128 tile has 1.33x the L2 -> LDS bytes
per flop of a 256 x 128 tile and 2x those of 256 x 256, so it cannot win on the plain forms; what the comparison shows is whether the
PENALTY of an epilogue form over the plain form shrinks when two blocks share a CU.
PAIRS=192 python tools/gemm_two_blocks_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"
M = int(os.environ.get("PAIRS", "192")) * 785
NB = 3


def timeit(fn, iters=12):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


FORMS = (("plain bf16 (proj dgrad)", 768, 768, {}), ("bf16 + bf16 residual (proj fwd, hybrid stream)", 768, 768, dict(res16=True)),
         ("fp32 + fp32 residual (proj fwd, fp32 streams)", 768, 768, dict(res32=True)), ("plain bf16 (qkv fwd)", 2304, 768, {}),
         ("QuickGELU + pre-activation (fc1 fwd)", 3072, 768, dict(act=True)), ("gate (fc2 dgrad)", 3072, 768, dict(gate=True)),
         ("bf16 + bf16 residual (fc2 fwd)", 768, 3072, dict(res16=True)), ("plain bf16 (fc1 dgrad)", 768, 3072, {}))
print(f"M = {M}; us per launch (TFLOP/s); penalty = epilogue form over the plain form of the same N x K, per kernel")
plain = {}
for name, n, k, kw in FORMS:
    As = [torch.randn(M, k, device=dev).bfloat16() for _ in range(NB)]
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    odt = torch.float32 if kw.get("res32") else torch.bfloat16
    Os = [torch.empty(M, n, dtype=odt, device=dev) for _ in range(NB)]
    extra = []
    for j in range(NB):
        e = {}
        if kw.get("res16"):
            e["residual"] = torch.randn(M, n, device=dev).bfloat16()
        if kw.get("res32"):
            e["residual"] = torch.randn(M, n, device=dev)
        if kw.get("act"):
            e.update(act="quick_gelu", preact=torch.empty(M, n, dtype=torch.bfloat16, device=dev))
        if kw.get("gate"):
            e.update(gate_h=torch.randn(M, n, device=dev).bfloat16(), gate_act="quick_gelu")
        extra.append(e)
    res = {}
    for tile in (256, 128):
        i = [0]

        def f():
            i[0] = (i[0] + 1) % NB
            K.gemm_nt(As[i[0]], w, Os[i[0]], bias=None if kw.get("gate") else bias, tile=tile, **extra[i[0]])
        res[tile] = sorted(timeit(f) for _ in range(3))[1]
    fl = 2.0 * M * n * k
    if not kw:
        plain[(n, k)] = dict(res)
    pen = ""
    if kw and (n, k) in plain:
        pen = f"   penalty 256: +{res[256] - plain[(n, k)][256]:5.0f} us   128 x 128, two per CU: +{res[128] - plain[(n, k)][128]:5.0f} us"
    print(f"{name:48s} N {n:4d} K {k:4d}: 256 x 256 {res[256]:6.1f} ({fl / res[256] / 1e6:5.0f})   128 x 128 {res[128]:6.1f} ({fl / res[128] / 1e6:5.0f}){pen}", flush=True)
    del As, Os, extra
