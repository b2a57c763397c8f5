"""e4m3 weight-gradient kernel against the bf16 one on the ViT-H/14 16-frame shapes (48 pairs: M = 58 416 token rows) and the B/16
ones (192 pairs: M = 150 720): microseconds (rotating buffers, medians) and TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, shapes in ((58416, [(3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120)]), (150720, [(2304, 768), (768, 768), (3072, 768), (768, 3072)])):
    for Na, Nb in shapes:
        Ps = [torch.randn(M, Na, device=dev).bfloat16() for _ in range(3)]
        Qs = [torch.randn(M, Nb, device=dev).bfloat16() for _ in range(3)]
        P8 = [K.quantize_fp8(p) for p in Ps]
        Q8 = [K.quantize_fp8(q) for q in Qs]
        out = torch.zeros(Na, Nb, device=dev)
        i = [0]

        def f16():
            i[0] = (i[0] + 1) % 3
            K.gemm_tn(Ps[i[0]], Qs[i[0]], out, accumulate=True)

        def f8():
            i[0] = (i[0] + 1) % 3
            K.gemm_tn_fp8(P8[i[0]][0], P8[i[0]][1], Q8[i[0]][0], Q8[i[0]][1], out, accumulate=True)
        t16 = sorted(timeit(f16) for _ in range(3))[1]
        t8 = sorted(timeit(f8) for _ in range(3))[1]
        fl = 2.0 * M * Na * Nb
        print(f"TN {M} x {Na} x {Nb}: bf16 {t16 * 1e3:7.1f} us {fl / t16 / 1e9:6.0f} TF | e4m3 {t8 * 1e3:7.1f} us {fl / t8 / 1e9:6.0f} TF | x{t16 / t8:.2f}", flush=True)
