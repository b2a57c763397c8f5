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
    cases = [("qkv fwd", 2304, 768, "plain"), ("proj f32+res", 768, 768, "res32"), ("fc1 dgrad", 768, 3072, "plain")]
    if os.environ.get("TRACE_WIDE"):  # round 6: the two-output / side-input forms whose epilogue is 8 - 11 us of a 31 us tile
        cases = [("fc1 fwd act", 3072, 768, "act"), ("fc2 dgrad gate", 3072, 768, "gate"), ("qkv fwd", 2304, 768, "plain")]
    for name, n, k, kind in cases:
        g = torch.Generator(device=dev).manual_seed(1)
        sets = []
        for _ in range(3):
            a = torch.randn(M, k, generator=g, device=dev).bfloat16()
            kw = dict(bias=torch.randn(n, generator=g, device=dev))
            odt = torch.bfloat16
            if kind == "res32":
                kw["residual"] = torch.randn(M, n, generator=g, device=dev); odt = torch.float32
            if kind == "act":
                kw.update(act="quick_gelu", preact=torch.empty(M, n, dtype=torch.bfloat16, device=dev))
            if kind == "gate":
                kw = dict(gate_h=torch.randn(M, n, generator=g, device=dev).bfloat16(), gate_act="quick_gelu")
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
            # does an epilogue take longer when more blocks are inside one at the same time?  (its length against the number of blocks inside an
            # epilogue at its midpoint, by quartile of that number: chip-wide write bandwidth shared by coinciding epilogues would show here)
            mids, lens = ((e0 + e1) / 2)[valid], (e1 - e0)[valid]
            conc = np.array([(valid & (e0 <= x) & (e1 >= x)).sum() for x in mids])
            qs = np.percentile(conc, [25, 50, 75])
            parts = [lens[conc <= qs[0]], lens[(conc > qs[0]) & (conc <= qs[1])], lens[(conc > qs[1]) & (conc <= qs[2])], lens[conc > qs[2]]]
            print(f"{'':14s} {'':8s}  epilogue length by concurrency quartile (blocks inside an epilogue at its midpoint <= {qs[0]:.0f} / <= {qs[1]:.0f} / <= {qs[2]:.0f} / more): "
                  + " / ".join(f"{p.mean():.2f}" if len(p) else "-" for p in parts) + f" us; correlation {np.corrcoef(conc, lens)[0, 1]:+.2f}", flush=True)


if __name__ == "__main__":
    main()
