#!/usr/bin/env python3
"""Per-tile time stamps of the 256x256 NT kernel (experiment library, ABL & 2048): where a block's time goes -- main loop vs
epilogue -- and how the blocks' epilogues are phased against each other (dev tool, GPU only).

    python experiments/gemm_trace.py [PAIRS=192]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import numpy as np  # noqa: E402
import gemm_ab as AB  # noqa: E402

AB.lib.tvts_exp_set_trace.argtypes = [ctypes.c_void_p]


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    M = pairs * 785
    dev = "cuda:0"
    trace = torch.zeros(256 * 32 * 6, dtype=torch.int64, device=dev)
    AB.lib.tvts_exp_set_trace(ctypes.c_void_p(trace.data_ptr()))
    for name, n, k, kind in [("qkv fwd", 2304, 768, "plain"), ("proj f32+res", 768, 768, "res32"), ("fc1 dgrad", 768, 3072, "plain")]:
        g = torch.Generator(device=dev).manual_seed(1)
        sets = []
        for _ in range(3):
            a = torch.randn(M, k, generator=g, device=dev).bfloat16()
            kw = dict(bias=torch.randn(n, generator=g, device=dev))
            odt = torch.bfloat16
            if kind == "res32":
                kw["residual"] = torch.randn(M, n, generator=g, device=dev); odt = torch.float32
            sets.append((a, kw, torch.empty(M, n, dtype=odt, device=dev)))
        b = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
        for vn, v in [("prod", 2058), ("stag4", 2074), ("reg+cnt", 3594)]:
            for rep in range(3):  # last repetition is reported (rotating sets: nothing cached from the previous launch)
                a, kw, out = sets[rep % 3]
                trace.zero_()
                AB.exp_gemm(v, -1, (0, 0), a, b, out, **kw)
                torch.cuda.synchronize()
            t = trace.cpu().numpy().reshape(256, 32, 6).astype(np.float64)
            valid = t[:, :, 0] > 0
            rt0 = t[:, :, 0][valid].min()
            e0, e1, e2 = (t[:, :, 0] - rt0) * 0.01, (t[:, :, 2] - rt0) * 0.01, (t[:, :, 4] - rt0) * 0.01  # us (100 MHz)
            epi = (e1 - e0)[valid]
            epi_all = (e2 - e0)[valid & (t[:, :, 4] > 0)]
            ntl = valid.sum(1)
            # tile period of a block: start of epilogue k+1 - start of epilogue k
            per = np.concatenate([np.diff(e0[bk, :ntl[bk]]) for bk in range(256) if ntl[bk] > 1])
            first = e0[:, 0][valid[:, 0]]
            last_end = np.array([e1[bk, ntl[bk] - 1] for bk in range(256) if ntl[bk] > 0])
            print(f"{name:14s} {vn:8s}: tiles/block {ntl.min()}-{ntl.max()}  tile period {per.mean():6.2f} us (p10 {np.percentile(per, 10):.2f} p90 {np.percentile(per, 90):.2f})"
                  f"  epilogue wave0 {epi.mean():5.2f} us (p10 {np.percentile(epi, 10):.2f} p90 {np.percentile(epi, 90):.2f})"
                  f"  epi start -> next barrier {epi_all.mean():5.2f} us  first-epilogue start spread {first.min():.1f}..{first.max():.1f} us"
                  f"  kernel end spread {last_end.min():.1f}..{last_end.max():.1f} us", flush=True)
            # how many blocks are inside an epilogue at a time: sample the time line
            ts = np.linspace(first.max(), last_end.min(), 400)
            busy = [(valid & (e0 <= x) & (e1 >= x)).sum() for x in ts]
            print(f"{'':14s} {'':8s}  blocks inside an epilogue (wave 0) over the steady part: mean {np.mean(busy):.1f} max {np.max(busy)} min {np.min(busy)} of 256", flush=True)


if __name__ == "__main__":
    main()
