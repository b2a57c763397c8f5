#!/usr/bin/env python3
"""Weight-gradient (TN) GEMM: the 128x128 kernel against the pipelined 256x256 kernel on the step's shapes, interleaved rounds,
results checked against the fp32 product (dev tool, GPU only).   python tools/tn_ab.py [PAIRS=192] [ROUNDS=5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tvts_amd import _lib, hip as K  # noqa: E402


def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.load()
    M, Mt = pairs * 785, pairs * 4 * 32
    dev = "cuda:0"
    cases = [("qkv wgrad", M, 2304, 768), ("fc1 wgrad", M, 3072, 768), ("fc2 wgrad", M, 768, 3072), ("proj wgrad", M, 768, 768),
             ("text fc", Mt, 2048, 512), ("text qkv", Mt, 1536, 512), ("text proj", Mt, 512, 512), ("ragged", 40001, 1280, 640)]
    tot = {128: 0.0, 256: 0.0}
    for name, m, na, nb in cases:
        g = torch.Generator(device=dev).manual_seed(na + nb)
        sets = [(torch.randn(m, na, generator=g, device=dev).bfloat16(), torch.randn(m, nb, generator=g, device=dev).bfloat16())
                for _ in range(3 if m > 100000 else 1)]
        p, q = sets[0]
        ref = (p.float().t() @ q.float())
        csr = p.float().sum(0)
        line = f"{name:11s} {m}x{na}x{nb}:"
        ts = {128: [], 256: [], "256nc": []}
        for tile in (128, 256):
            out = torch.full((na, nb), float("nan"), device=dev)
            cs = torch.zeros(na, device=dev)
            K.gemm_tn(p, q, out, accumulate=False, colsum=cs, tile=tile)
            err = float((out - ref).norm() / ref.norm())
            amax = float((out - ref).abs().max() / ref.abs().max())
            cerr = float((cs - csr).abs().max() / csr.abs().max())
            ok = err < 3e-5 and amax < 1e-3 and cerr < 1e-3
            line += f" | {tile}: {'ok' if ok else 'WRONG'} rel {err:.1e} max {amax:.1e} cs {cerr:.1e}"
        outs = [torch.empty(na, nb, device=dev) for _ in sets]
        css = torch.zeros(na, device=dev)
        for _ in range(rounds):
            for tile in (128, 256):
                def run(tile=tile):
                    for (pp, qq), oo in zip(sets, outs):
                        K.gemm_tn(pp, qq, oo, accumulate=False, colsum=css, tile=tile)
                ts[tile].append(timeit(run) / len(sets))

            def run_nc():  # the 256 kernel without the bias-gradient column sums
                for (pp, qq), oo in zip(sets, outs):
                    K.gemm_tn(pp, qq, oo, accumulate=False, colsum=None, tile=256)
            ts["256nc"].append(timeit(run_nc) / len(sets))
        for tile in (128, 256, "256nc"):
            med = sorted(ts[tile])[len(ts[tile]) // 2]
            tot[tile] = tot.get(tile, 0.0) + med
            line += f" | {tile} {med * 1e3:7.1f}us {2.0 * m * na * nb / med / 1e9:5.0f}TF"
        line += f" | auto picks {K.gemm_tn_select(m, na, nb)}"
        print(line, flush=True)
    print("sum of medians (ms):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
