#!/usr/bin/env python3
"""The weight-gradient kernel against the number of contraction ranges (TVTS_TN_SPLITS in the call's opts): round efficiency vs L2 locality
(with 8 ranges every XCD owns exactly one; the automatic choice fills the round).  GPU only.  usage: tn_splits.py [PAIRS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import _lib, hip as K  # noqa: E402

dev = "cuda:0"
M = int(sys.argv[1]) * 785 if len(sys.argv) > 1 else 192 * 785


def timeit(fn, iters=6):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


lib = _lib.load()
for na, nb in ((2304, 768), (3072, 768), (768, 3072), (768, 768)):
    ps = [torch.randn(M, na, device=dev).bfloat16() for _ in range(3)]
    qs = [torch.randn(M, nb, device=dev).bfloat16() for _ in range(3)]
    out = torch.zeros(na, nb, device=dev)
    cs = torch.zeros(na, device=dev)
    line = f"TN {M} x {na} x {nb}:"
    for sp in (0, 8, 16, 24, 32):
        i = [0]

        def f():
            i[0] = (i[0] + 1) % 3
            K.gemm_tn(ps[i[0]], qs[i[0]], out, accumulate=True, colsum=cs, splits=sp)
        ms = timeit(f)
        line += f"  s{sp}: {ms * 1e3:6.1f}us {2.0 * M * na * nb / ms / 1e9:5.0f}TF"
    print(line, flush=True)
