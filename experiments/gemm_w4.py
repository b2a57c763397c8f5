#!/usr/bin/env python3
"""EXPERIMENT: the four-wave (one per SIMD, 128x128 wave tile, AGPR accumulators) 256x256 NT kernel of experiments/csrc/gemm_w4.hip
against the production kernel (plain bf16 output both) and hipBLASLt.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC
-I include -shared experiments/csrc/gemm_w4.hip -o tvts_amd/libtvts_w4.so   (dev tool, GPU only; delete the .so afterwards)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "experiments", "libtvts_w4.so"))
ci, vp = ctypes.c_int, ctypes.c_void_p
lib.tvts_exp_gemm_w4.argtypes = [ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp]
lib.tvts_exp_gemm_w4.restype = ci


def w4(pin, a, b, out):
    rc = lib.tvts_exp_gemm_w4(pin, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), a.shape[0], b.shape[0], a.shape[1],
                              out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M = 128 * 785
for (m, n, k) in ((1000, 512, 256), (4096, 4096, 4096), (8192, 8192, 8192), (M, 768, 3072), (M, 768, 2304), (M, 2304, 768), (M, 3072, 768), (M, 768, 768)):
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    ref = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    K.gemm_nt(a, b, ref)
    line = f"{m}x{n}x{k}:"
    for pin in (0, 1, 2):
        out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        w4(pin, a, b, out)
        torch.cuda.synchronize()
        err = float((out.float() - ref.float()).norm() / ref.float().norm())
        ms = timeit(lambda: w4(pin, a, b, out))
        tag = ("", "p", "i")[pin]
        line += f" w4{tag} {'ok' if err < 1e-3 else 'WRONG %.1e' % err} {2.0 * m * n * k / ms / 1e9:6.0f} TF |"
    ms = timeit(lambda: K.gemm_nt(a, b, ref))
    ms2 = timeit(lambda: torch.matmul(a, b.t(), out=ref))
    print(line + f" production {2.0 * m * n * k / ms / 1e9:6.0f} TF | hipBLASLt {2.0 * m * n * k / ms2 / 1e9:6.0f} TF", flush=True)
