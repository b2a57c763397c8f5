"""Stream-K against the tile-granular walk of the NT kernels at the reference's per-GPU batches (M = pairs x 785 ViT rows, pairs x 128
text rows, pairs x 789 sort-head rows): time (rotating buffers, medians), error against an fp64 product, and that two stream-K
launches give the same bits.  PAIRS="2 6 12 24 48" selects the batches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def med(fn):
    return sorted(timeit(fn) for _ in range(3))[1]


pairs_list = [int(x) for x in os.environ.get("PAIRS", "2 6 12 24 48").split()]
for pairs in pairs_list:
    for (M, shapes) in ((pairs * 785, [(768, 768), (2304, 768), (3072, 768), (768, 2304), (768, 3072)]),
                        (pairs * 128, [(512, 512), (1536, 512), (2048, 512), (512, 2048)]),
                        (pairs * 789, [(512, 512), (1536, 512), (2048, 512), (512, 2048)])):
        for (n, k) in shapes:
            As = [torch.randn(M, k, device=dev).bfloat16() for _ in range(4)]
            b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
            bias = torch.randn(n, device=dev)
            Os = [torch.empty(M, n, dtype=torch.bfloat16, device=dev) for _ in range(4)]
            res = {}
            for name, kw in (("128", dict(tile=128, streamk=False)), ("256", dict(tile=256, streamk=False)), ("auto", dict()),
                             ("sk", dict(streamk=True))):
                i = [0]

                def f():
                    i[0] = (i[0] + 1) % 4
                    K.gemm_nt(As[i[0]], b, Os[i[0]], bias=bias, **kw)
                try:
                    res[name] = med(f)
                except K.HipError:
                    res[name] = float("nan")
            line = f"pairs {pairs:3d} M {M:6d} N {n:5d} K {k:5d}: " + " | ".join(f"{nm} {res[nm] * 1e3:6.1f}" for nm in res)
            if res["sk"] == res["sk"]:
                ref = (As[0].double() @ b.double().t() + bias.double())
                o1 = torch.empty(M, n, dtype=torch.float32, device=dev)
                o2 = torch.empty_like(o1)
                o3 = torch.empty_like(o1)
                K.gemm_nt(As[0], b, o1, bias=bias, streamk=True)
                K.gemm_nt(As[1], b, o3, bias=bias, streamk=True)  # another launch in between
                K.gemm_nt(As[0], b, o2, bias=bias, streamk=True)
                K.gemm_nt(As[0], b, o3, bias=bias, tile=256, streamk=False)
                torch.cuda.synchronize()
                e_sk = float((o1.double() - ref).abs().max()); e_dp = float((o3.double() - ref).abs().max())
                best = min(v for kname, v in res.items() if kname in ("128", "256") and v == v)
                line += f" | sk/best {res['sk'] / best:5.2f} auto/best {res['auto'] / best:5.2f} | err sk {e_sk:.2e} dp {e_dp:.2e} same bits {bool((o1 == o2).all())}"
            print(line, flush=True)
