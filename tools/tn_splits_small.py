#!/usr/bin/env python3
"""Weight-gradient GEMMs at the reference's per-GPU batches (12 / 24 pairs: M = 9 420 / 18 840 token rows, text tower 1 536 / 3 072):
kernel + reduce time against the number of contraction ranges, both tile sizes; the automatic choice marked.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=30):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for pairs in (12, 24):
    M = pairs * 785
    for na, nb, m in ((2304, 768, M), (3072, 768, M), (768, 3072, M), (768, 768, M), (1536, 512, pairs * 128), (2048, 512, pairs * 128)):
        ps = [torch.randn(m, na, device=dev).bfloat16() for _ in range(3)]
        qs = [torch.randn(m, nb, device=dev).bfloat16() for _ in range(3)]
        out = torch.zeros(na, nb, device=dev)
        cs = torch.zeros(na, device=dev)
        line = f"pairs {pairs:2d} TN {m:6d} x {na:4d} x {nb:4d}:"
        for tile in (128, 256):
            best = None
            for sp in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28):
                if sp > 1 and m // sp < 768:
                    continue
                i = [0]

                def f():
                    i[0] = (i[0] + 1) % 3
                    K.gemm_tn(ps[i[0]], qs[i[0]], out, accumulate=True, colsum=cs, splits=sp, tile=tile)
                t = timeit(f)
                if sp == 0:
                    auto = t
                elif best is None or t < best[0]:
                    best = (t, sp)
            line += f" | {tile}: auto {auto:6.1f} us, best {best[0]:6.1f} us at {best[1]:2d} ranges"
        print(line + f" | auto picks {K.gemm_tn_select(m, na, nb)}", flush=True)
