#!/usr/bin/env python3
"""A/B of NT GEMM kernel variants from the bench-only experiment library (experiments/csrc -> libtvts_exp.so) against the
production kernel on the B/16 step's shapes: interleaved rounds in one process, medians (dev tool, GPU only).

    python experiments/gemm_ab.py [PAIRS=192] [ROUNDS=7]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "experiments", "libtvts_exp.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.tvts_exp_gemm_nt.argtypes = [ci, ci, ci, ci, vp, ci, vp, ci, ci, ci, ci, vp, vp, ci, ci, vp, ci, vp, ci, ci, vp, ci, ci, vp]
lib.tvts_exp_gemm_nt.restype = ci
ACT = {"": 0, "quick_gelu": 1, "gelu": 2, "add": 3}


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def exp_gemm(variant, gc, stag, a, b, out, bias=None, residual=None, act="", preact=None, gate_h=None, gate_act=""):
    M, Kd = a.shape
    N = b.shape[0]
    rc = lib.tvts_exp_gemm_nt(variant, gc, stag[0], stag[1], P(a), a.stride(0), P(b), b.stride(0), M, N, Kd, P(bias), P(residual),
                              residual.stride(0) if residual is not None else 0, ACT[act], P(preact),
                              preact.stride(0) if preact is not None else 0, P(gate_h), gate_h.stride(0) if gate_h is not None else 0,
                              ACT[gate_act], P(out), out.stride(0), 1 if out.dtype == torch.float32 else 0,
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    M = pairs * 785
    dev = "cuda:0"
    cases = [  # (name, M, N, K, kind)
        ("qkv fwd", M, 2304, 768, "plain"), ("proj f32+res", M, 768, 768, "res32"), ("proj bf16+res", M, 768, 768, "res16"), ("proj bf16", M, 768, 768, "plain"),
        ("proj bf16+bf16res", M, 768, 768, "add16"), ("fc2 fwd bf16+bf16res", M, 768, 3072, "add16"),
        ("fc1 fwd gelu+pre", M, 3072, 768, "act"), ("fc2 dgrad gate", M, 3072, 768, "gate"), ("fc2 fwd f32+res", M, 768, 3072, "res32"),
        ("fc1 dgrad", M, 768, 3072, "plain"), ("qkv dgrad", M, 768, 2304, "plain"), ("square", 4096, 4096, 4096, "plain")]
    # variant 10 + ABL = the production kernel with a compile-time epilogue ablation (results are wrong by construction, not checked):
    # 11 no epilogue, 12 no side-input loads (residual / gate), 14 no stores, 16 neither loads nor stores (LDS transpose + math only)
    catalog = {"prod": 0, "no-epi": 11, "no-side-loads": 12, "no-stores": 14, "lds+math only": 16, "stag2": 18, "stag4": 26, "sc1 st": 42,
               "plain st": 74, "nt side ld": 138, "side prefetch": 266, "prefetch+nt ld": 394, "cnt vmcnt": 522, "reg": 1034,
               "reg+cnt": 1546, "reg+cnt+stag4": 1562, "m32": 1, "pasm": 8202, "pasm+cnt": 8714, "stagdma": 32778, "pasm+stagdma": 40970, "no-epi+stagdma": 32779,
               "pin": 163850, "pasm+pin": 172042, "no-epi+pin": 163851,
               "b16 generic": 8552458, "b16 pasm": 8560650, "b16 side": 25337866}  # round 6: bf16-first patch (production flags + ABL 8388608)
    names = os.environ.get("AB_VARIANTS", "prod,side prefetch,no-side-loads").split(",")
    variants = [(n, catalog[n], -1, (0, 0)) for n in names]
    tot = {v[0]: 0.0 for v in variants}
    # NSETS independent operand / output sets used round-robin inside the timed loop: in the training step a GEMM's activations
    # were written by the previous kernel and its output is read by the next one -- nothing is re-read from one launch to the
    # next.  With ONE buffer set the 231 MB activation operand stays in the 256 MiB Infinity Cache from iteration to iteration
    # unless the output stream evicts it, which made store-policy variants look 20 % apart that are equal in the step.
    nsets = int(os.environ.get("AB_SETS", "3"))
    for name, m, n, k, kind in cases:
        g = torch.Generator(device=dev).manual_seed(n + k)
        sets = []
        for si in range(nsets if m > 10000 else 1):
            a = torch.randn(m, k, generator=g, device=dev).bfloat16()
            kw = dict(bias=torch.randn(n, generator=g, device=dev))
            odt = torch.bfloat16
            if kind == "res32":
                kw["residual"] = torch.randn(m, n, generator=g, device=dev); odt = torch.float32
            elif kind == "res16":
                kw["residual"] = torch.randn(m, n, generator=g, device=dev)
            elif kind == "act":
                kw.update(act="quick_gelu", preact=torch.empty(m, n, dtype=torch.bfloat16, device=dev))
            elif kind == "gate":
                kw = dict(gate_h=torch.randn(m, n, generator=g, device=dev).bfloat16(), gate_act="quick_gelu")
            elif kind == "add16":  # the hybrid stream's forms: bf16 result + bf16 residual through the gate slot
                kw = dict(bias=kw["bias"], gate_h=torch.randn(m, n, generator=g, device=dev).bfloat16(), gate_act="add")
            sets.append((a, kw, torch.empty(m, n, dtype=odt, device=dev)))
        b = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
        a, kw, out = sets[0]
        ref = torch.empty(m, n, dtype=odt, device=dev)
        exp_gemm(0, -1, (0, 0), a, b, ref, **kw)
        extra = [("prod gc3", 0, 3, (0, 0))] if n == 2304 else []
        extra += [("prod gc0", 0, 0, (0, 0))] if n == 3072 else []
        vs = variants + extra
        for vn, v, gc, stag in vs:  # correctness of every variant against the production kernel's output
            if v in (11, 12, 14, 16, 32779, 163851):
                continue
            out.fill_(float("nan"))
            exp_gemm(v, gc, stag, a, b, out, **kw)
            d = (out.float() - ref.float())
            err, amax = float(d.norm() / ref.float().norm()), float(d.abs().max())  # NaN if anything was left unwritten
            ok = err < (3e-3 if odt == torch.bfloat16 else 1e-5) and amax <= (0.13 if odt == torch.bfloat16 else 2e-4)
            if not ok:
                print(f"  WRONG {name} / {vn}: rel {err:.3e} max-abs {amax:.3e}", flush=True)
            elif vn.startswith("b16"):  # the bf16-first patch claims the SAME bits as the fp32 patch (one rounding either way)
                if vn in ("b16 pasm", "b16 side"):  # the hand-scheduled kernels add the bias inside the K loop: compare with THEIR fp32-patch form
                    ref2 = torch.empty_like(ref)
                    try:
                        exp_gemm(catalog["pasm+pin"], gc, stag, a, b, ref2, **kw)
                        print(f"  {name} / {vn}: {'bit-identical to' if torch.equal(out, ref2) else 'DIFFERS from'} the hand-scheduled fp32-patch kernel", flush=True)
                    except AssertionError:
                        pass
                same = torch.equal(out, ref)
                pre_same = ""
                if kw.get("preact") is not None:
                    p_new = kw["preact"].clone()
                    exp_gemm(0, -1, (0, 0), a, b, ref, **kw)
                    pre_same = f", side output {'identical' if torch.equal(p_new, kw['preact']) else 'DIFFERS'}"
                print(f"  {name} / {vn}: result {'bit-identical to' if same else 'DIFFERS from'} the production kernel's (max-abs {amax:.3e}){pre_same}", flush=True)
        ts = {vn: [] for vn, *_ in vs}
        def run_sets(v, gc, stag):
            for a_, kw_, out_ in sets:
                exp_gemm(v, gc, stag, a_, b, out_, **kw_)
        for _ in range(rounds):
            for vn, v, gc, stag in vs:
                ts[vn].append(timeit(lambda: run_sets(v, gc, stag), iters=4) / len(sets))
        fl = 2.0 * m * n * k
        line = f"{name:18s} {m}x{n}x{k}:"
        for vn, *_ in vs:
            if not ts[vn]:
                continue
            med = sorted(ts[vn])[len(ts[vn]) // 2]
            if vn in tot:
                tot[vn] += med
            line += f" | {vn} {med * 1e3:7.1f}us {fl / med / 1e9:5.0f}TF"
        print(line, flush=True)
    print("sum of medians (ms):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
