#!/usr/bin/env python3
"""What does a CU's K loop wait for?  The 256x256 NT kernel (bf16 and e4m3 / scaled MFMA) on a FIXED number of tiles per CU with
8 .. 256 CUs streaming: if the per-CU rate is flat the loop is bound inside the CU (latency, LDS, matrix pipe); if it falls as
CUs are added the shared L2 / fabric fill is what the stage period waits for.  GPU only.  usage: gemm_cus.py [N K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import _lib, hip as K  # noqa: E402

if os.environ.get("TVTS_LIB"):  # dev: an experiment build of the kernel library (e.g. -DTVTS_LOOP_ABL=n timing ablations)
    _lib.LIB_PATH = os.path.abspath(os.environ["TVTS_LIB"])

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
Kd = int(sys.argv[2]) if len(sys.argv) > 2 else 768
TILES_PER_CU = 7


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
print(f"N={N} K={Kd}, {TILES_PER_CU} tiles per CU; TF per CU (bf16 | fp8 scaled MFMA), stage period from the K loop only is not separated")
for cus in [int(c) for c in os.environ.get("CUS", "8,32,64,128,192,256").split(",")]:
    tiles_m = cus * TILES_PER_CU // (N // 256)
    M = tiles_m * 256
    a = torch.randn(M, Kd, device=dev).bfloat16()
    b = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_bf = timeit(lambda: K.gemm_nt(a, b, out, tile=256, cus=cus))
    line = f"CUs {cus:3d}  M {M:6d}  bf16 {t_bf * 1e3:7.1f} us {2.0 * M * N * Kd / t_bf / 1e9 / cus:6.2f} TF/CU"
    if Kd % 128 == 0:
        a8, rs = K.quantize_fp8_rows(a)
        b8, sb = K.quantize_fp8(b)
        t_f8 = timeit(lambda: K.gemm_nt_fp8(a8, rs, b8, sb, out, cus=cus))
        line += f" | fp8 {t_f8 * 1e3:7.1f} us {2.0 * M * N * Kd / t_f8 / 1e9 / cus:6.2f} TF/CU"
    print(line, flush=True)
