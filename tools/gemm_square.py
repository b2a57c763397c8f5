#!/usr/bin/env python3
"""NT GEMM on square shapes (the guide's reference points: 4096^3 and 8192^3, uniform random operands)."""
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
    return e0.elapsed_time(e1) / iters

for n in (4096, 8192):
    a = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(n, n, dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: K.gemm_nt(a, b, out))
    print(f"NT {n}^3: {ms*1e3:8.1f} us  {2.0*n**3/ms/1e9:7.1f} TF")
    ms = timeit(lambda: torch.matmul(a, b.t(), out=out))
    print(f"hipBLASLt (torch.matmul) {n}^3: {ms*1e3:8.1f} us  {2.0*n**3/ms/1e9:7.1f} TF")

print("step shapes (M=100480): ours vs hipBLASLt reference")
M = 128 * 785
for (n, k) in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
    a = (torch.rand(M, k, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: K.gemm_nt(a, b, out))
    ms2 = timeit(lambda: torch.matmul(a, b.t(), out=out))
    fl = 2.0 * M * n * k
    print(f"NT {M}x{n}x{k}: ours {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF | hipBLASLt {ms2*1e3:7.1f} us {fl/ms2/1e9:7.1f} TF")
for (na, nb) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    p = (torch.rand(M, na, device="cuda") * 2 - 1).bfloat16()
    q = (torch.rand(M, nb, device="cuda") * 2 - 1).bfloat16()
    out = torch.zeros(na, nb, device="cuda")
    outb = torch.zeros(na, nb, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm_tn(p, q, out, accumulate=True))
    ms2 = timeit(lambda: torch.matmul(p.t(), q, out=outb))
    fl = 2.0 * M * na * nb
    print(f"TN {M}x{na}x{nb}: ours {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF | hipBLASLt {ms2*1e3:7.1f} us {fl/ms2/1e9:7.1f} TF")

print("fp8 (e4m3, per-tensor scales) on the same pipelined kernel")
for (m, n, k) in ((4096, 4096, 4096), (8192, 8192, 8192), (M, 2304, 768), (M, 3072, 768), (M, 768, 3072), (38944, 3840, 1280), (38944, 5120, 1280), (38944, 1280, 5120)):
    a = (torch.rand(m, k, device="cuda") * 2 - 1)
    b = (torch.rand(n, k, device="cuda") * 2 - 1)
    a8, sa = K.quantize_fp8(a.bfloat16()); b8, sb = K.quantize_fp8(b.bfloat16())
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: K.gemm_nt_fp8(a8, sa, b8, sb, out))
    ab, bb = a.bfloat16(), b.bfloat16()
    ms2 = timeit(lambda: K.gemm_nt(ab, bb, out))
    fl = 2.0 * m * n * k
    print(f"NT {m}x{n}x{k}: fp8 {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF | bf16 {ms2*1e3:7.1f} us {fl/ms2/1e9:7.1f} TF")
