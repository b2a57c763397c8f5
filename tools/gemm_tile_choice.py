"""128x128 vs 256x256 NT kernel at the small row counts of the reference's per-GPU batches (M = pairs x 785, text tower M = pairs x 128):
where the automatic tile choice (>= NT_MIN_TILES_256 output tiles of 256 x 256) should sit.  Rotating buffers, medians."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for pairs in (2, 6, 12, 24, 48):
    for (M, shapes) in ((pairs * 785, [(768, 768), (2304, 768), (3072, 768), (768, 2304), (768, 3072)]),
                        (pairs * 128, [(512, 512), (1536, 512), (2048, 512), (512, 2048)]),
                        (pairs * 789, [(512, 512), (1536, 512), (2048, 512), (512, 2048)])):
        for (n, k) in shapes:
            As = [torch.randn(M, k, device=dev).bfloat16() for _ in range(4)]
            b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
            Os = [torch.empty(M, n, dtype=torch.bfloat16, device=dev) for _ in range(4)]
            res = {}
            for tile in (128, 256):
                i = [0]

                def f():
                    i[0] = (i[0] + 1) % 4
                    K.gemm_nt(As[i[0]], b, Os[i[0]], tile=tile)
                res[tile] = sorted(timeit(f) for _ in range(3))[1]
            t256 = -(-M // 256) * (n // 256)
            auto = K.gemm_nt_select(M, n)
            best = 128 if res[128] < res[256] else 256
            print(f"pairs {pairs:3d} M {M:6d} N {n:5d} K {k:5d}: 256-tiles {t256:4d} | 128: {res[128] * 1e3:6.1f} us | 256: {res[256] * 1e3:6.1f} us | auto {auto} best {best}"
                  f"{'  <-- auto is not best (%.0f %%)' % (100 * (res[auto] / res[best] - 1)) if auto != best else ''}", flush=True)
