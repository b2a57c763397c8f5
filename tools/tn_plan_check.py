#!/usr/bin/env python3
"""The weight-gradient entry point's automatic plan (tile, contraction ranges: the cost model in csrc/gemm.hip) against both tiles
with their own best range counts, on the step's shapes at several batch sizes.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


worst = 0.0
for pairs in (2, 12, 24, 48, 192):
    M, Mt, Ms = pairs * 785, pairs * 128, pairs * 789
    for m, na, nb in ((M, 2304, 768), (M, 3072, 768), (M, 768, 3072), (M, 768, 768), (Mt, 2048, 512), (Mt, 512, 2048), (Mt, 1536, 512),
                      (Mt, 512, 512), (Ms, 2048, 512), (Ms, 1536, 512), (Ms, 512, 512)):
        nset = 3 if m * (na + nb) * 2 < 400e6 else 1
        ps = [torch.randn(m, na, device=dev).bfloat16() for _ in range(nset)]
        qs = [torch.randn(m, nb, device=dev).bfloat16() for _ in range(nset)]
        out = torch.zeros(na, nb, device=dev)
        cs = torch.zeros(na, device=dev)
        res = {}
        for tile in (128, 256, None):
            best = None
            for sp in ((0,) if tile is None else (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32)):
                if sp > 1 and m // sp < 768:
                    continue
                i = [0]

                def f():
                    i[0] = (i[0] + 1) % nset
                    K.gemm_tn(ps[i[0]], qs[i[0]], out, accumulate=True, colsum=cs, splits=sp, tile=tile)
                t = timeit(f)
                if best is None or t < best[0]:
                    best = (t, sp)
            res[tile] = best
        b = min(res[128][0], res[256][0])
        off = res[None][0] / b - 1.0
        worst = max(worst, off)
        print(f"pairs {pairs:3d} TN {m:6d} x {na:4d} x {nb:4d}: auto ({K.gemm_tn_select(m, na, nb)}) {res[None][0]:7.1f} us | 128 best {res[128][0]:7.1f} us at "
              f"{res[128][1]:2d} | 256 best {res[256][0]:7.1f} us at {res[256][1]:2d} | auto is {100 * off:+5.1f} % off the best", flush=True)
print(f"worst: {100 * worst:.1f} %")
