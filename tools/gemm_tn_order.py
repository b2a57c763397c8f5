#!/usr/bin/env python3
"""TN (weight-gradient) kernel: tile walk order inside an m-range (TVTS_TN_AFAST dev knob), interleaved medians.  GPU only."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"
libc = ctypes.CDLL(None)


def timeit(fn, iters=8):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, na, nb in ((150720, 768, 3072), (150720, 3072, 768), (150720, 2304, 768), (150720, 768, 768),
                  (59184, 1280, 5120), (59184, 5120, 1280), (59184, 3840, 1280), (59184, 1280, 1280)):
    p = torch.randn(M, na, device=dev).bfloat16()
    q = torch.randn(M, nb, device=dev).bfloat16()
    out = torch.empty(na, nb, device=dev)
    cs = torch.empty(na, device=dev)
    ref = None
    times = {0: [], 1: []}
    ok = True
    for rnd in range(7):
        for v in (0, 1):
            libc.setenv(os.environ.get("KNOB", "TVTS_TN_AFAST").encode(), str(v).encode(), 1)
            times[v].append(timeit(lambda: K.gemm_tn(p, q, out, colsum=cs, accumulate=False)))
            if ref is None:
                ref = (out.clone(), cs.clone())
            ok = ok and torch.equal(ref[0], out)
    libc.unsetenv(os.environ.get("KNOB", "TVTS_TN_AFAST").encode())
    line = f"M={M} Na={na:5d} Nb={nb:5d}"
    for v in (0, 1):
        ms = sorted(times[v])[3]
        line += f" | {os.environ.get('KNOB', 'TVTS_TN_AFAST')}={v}: {ms * 1e3:6.1f}us {2.0 * M * na * nb / ms / 1e9:5.0f}TF"
    print(line, "" if ok else "MISMATCH", flush=True)
