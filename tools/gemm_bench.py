#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM entry points on the shapes of the B/16 step (dev tool, GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

B = int(os.environ.get("PAIRS", "64"))
M = B * 785
NT_SHAPES = [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (M, 768, 2304), (M, 512, 768), (B * 789, 2048, 512),
             (B * 4 * 32, 1536, 512)]
TN_SHAPES = [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (B * 789, 2048, 512)]


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


def main():
    dev = "cuda:0"
    from tvts_amd import _lib
    for tile in [int(x) for x in os.environ.get('TILES', '0,128,256').split(',')]:
        print(f"--- NT tile {tile}")
        from tvts_amd import hip
        with hip.options(nt_tile=tile):
            nt(dev)
    tn(dev)


def nt(dev):
    tot_f, tot_t = 0.0, 0.0
    for (m, n, k) in NT_SHAPES:
        a = torch.randn(m, k, device=dev).bfloat16()
        b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
        bias = torch.randn(n, device=dev)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        K.gemm_nt(a, b, out, bias=bias)
        ref = (a[-512:].float() @ b.float().t() + bias)
        err = float((out[-512:].float() - ref).norm() / ref.norm())
        ms = timeit(lambda: K.gemm_nt(a, b, out, bias=bias))
        fl = 2.0 * m * n * k
        tot_f += fl; tot_t += ms
        print(f"NT {m:6d} x {n:5d} x {k:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF  relerr {err:.1e}")
    print(f"NT total {tot_f/tot_t/1e9:.1f} TF")


def tn(dev):
    tot_f, tot_t = 0.0, 0.0
    for (m, na, nb) in TN_SHAPES:
        p = torch.randn(m, na, device=dev).bfloat16()
        q = torch.randn(m, nb, device=dev).bfloat16()
        out = torch.zeros(na, nb, device=dev)
        cs = torch.zeros(na, device=dev)
        K.gemm_tn(p, q, out, accumulate=False, colsum=cs)
        ref = p[:, :256].float().t() @ q.float()
        err = float((out[:256] - ref).norm() / ref.norm())
        ms = timeit(lambda: K.gemm_tn(p, q, out, accumulate=True, colsum=cs))
        fl = 2.0 * m * na * nb
        tot_f += fl; tot_t += ms
        print(f"TN {m:6d} x {na:5d} x {nb:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF  relerr {err:.1e}")
    print(f"TN total {tot_f/tot_t/1e9:.1f} TF")


if __name__ == "__main__":
    main()
