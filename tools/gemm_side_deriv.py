"""The MLP's two epilogue-heavy GEMMs with the saved tensor as pre-activation (rounds 1-4) and as act'(pre-activation) (round 5,
TVTS_GEMM_SIDE_DERIV) at the bench's row count, rotating buffers.  PAIRS=192 python tools/gemm_side_deriv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"
NB = 3


def timeit(fn, iters=12):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for label, rows, W, act in (("B/16 QuickGELU", int(os.environ.get("PAIRS", "192")) * 785, 768, "quick_gelu"), ("H/14 erf-GELU", 48 * 1217, 1280, "gelu")):
    M = rows
    x = [torch.randn(M, W, device=dev).bfloat16() for _ in range(NB)]
    dy = [torch.randn(M, W, device=dev).bfloat16() for _ in range(NB)]
    w1 = (torch.randn(4 * W, W, device=dev) * W ** -0.5).bfloat16()
    w2t = (torch.randn(4 * W, W, device=dev) * W ** -0.5).bfloat16()   # [4W, W]: the transposed fc2 weight, operand of the input gradient
    bias = torch.randn(4 * W, device=dev)
    a = [torch.empty(M, 4 * W, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    h = [torch.empty(M, 4 * W, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    dh = [torch.empty(M, 4 * W, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    for deriv in (False, True):
        i = [0]

        def fwd():
            i[0] = (i[0] + 1) % NB
            K.gemm_nt(x[i[0]], w1, a[i[0]], bias=bias, act=act, preact=h[i[0]], side_deriv=deriv)

        def bwd():
            i[0] = (i[0] + 1) % NB
            K.gemm_nt(dy[i[0]], w2t, dh[i[0]], gate_h=h[i[0]], gate_act=act, side_deriv=deriv)
        tf = sorted(timeit(fwd) for _ in range(3))[1]
        tb = sorted(timeit(bwd) for _ in range(3))[1]
        fl = 2.0 * M * W * 4 * W
        print(f"{label:16s} M {M:6d} saved tensor = {'derivative    ' if deriv else 'pre-activation'}: fc1 forward {tf:6.1f} us ({fl / tf / 1e6:5.0f} TF)   fc2 input gradient {tb:6.1f} us ({fl / tb / 1e6:5.0f} TF)", flush=True)
