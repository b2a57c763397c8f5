"""The ring form of the 128 x 128 NT kernel against the double-buffered 128 and the pipelined 256 kernel on the shapes of the
step at the reference's 12 / 24 pairs per GPU (rotating buffer sets; results compared bit for bit with the 128 kernel)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tvts_amd import hip as K
dev = "cuda:0"
def timeit(fn, calls=24, reps=5):
    """one hipGraph of `calls` launches, replayed: kernel time without the host's launch cost (12 us per call from Python)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(calls): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): gr.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * calls) * 1e3
pairs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,24").split(",")]
for P in pairs:
    Mv, Mt = P * 8 * 60 + P * 8, P * 4 * 32  # video rows (59 kept patches + CLS per frame, + per-clip CLS rows) / text rows
    shapes = [("qkv", Mv, 2304, 768, {}), ("proj+res", Mv, 768, 768, dict(res=True, f32=True)), ("fc1 gelu", Mv, 3072, 768, dict(act="quick_gelu", preact=True)),
              ("fc2+res", Mv, 768, 3072, dict(res=True, f32=True)), ("fc2 dgrad gate", Mv, 3072, 768, dict(gate=True)), ("fc1 dgrad", Mv, 768, 3072, {}),
              ("qkv dgrad", Mv, 768, 2304, {}), ("text qkv", Mt, 1536, 512, {}), ("text proj", Mt, 512, 512, dict(res=True, f32=True)),
              ("text fc1", Mt, 2048, 512, dict(act="quick_gelu", preact=True)), ("text fc2", Mt, 512, 2048, dict(res=True, f32=True))]
    print(f"--- {P} pairs per GPU (video rows {Mv}, text rows {Mt})")
    for name, m, n, k, kw in shapes:
        R = int(os.environ.get('RING_R', '3'))
        A = [torch.randn(m, k, device=dev).bfloat16() for _ in range(R)]
        B = [(torch.randn(n, k, device=dev) * k ** -0.5).bfloat16() for _ in range(R)]
        bias = torch.randn(n, device=dev)
        odt = torch.float32 if kw.get("f32") else torch.bfloat16
        O = [torch.empty(m, n, dtype=odt, device=dev) for _ in range(R)]
        res = [torch.randn(m, n, device=dev) for _ in range(R)] if kw.get("res") else [None] * R
        pre = [torch.empty(m, n, dtype=torch.bfloat16, device=dev) for _ in range(R)] if kw.get("preact") else [None] * R
        gh = [torch.randn(m, n, device=dev).bfloat16() for _ in range(R)] if kw.get("gate") else [None] * R
        line = f"{name:15s} {m:6d}x{n:5d}x{k:5d}:"
        ref = None
        for tile in (128, 256, "ring2", "ring3", "ring4"):
            i = [0]
            def f():
                i[0] = (i[0] + 1) % R
                j = i[0]
                K.gemm_nt(A[j], B[j], O[j], bias=bias, residual=res[j], act=kw.get("act"), preact=pre[j], gate_h=gh[j],
                          gate_act="quick_gelu" if kw.get("gate") else None, tile=tile)
            try:
                us = timeit(f)
            except Exception as e:
                line += f"  {tile}: n/a"
                continue
            K.gemm_nt(A[0], B[0], O[0], bias=bias, residual=res[0], act=kw.get("act"), preact=pre[0], gate_h=gh[0],
                      gate_act="quick_gelu" if kw.get("gate") else None, tile=tile)
            got = O[0].clone()
            if tile == 128: ref = got
            same = "" if tile in (128, 256) else (" =" if torch.equal(got, ref) else " DIFF")
            line += f"  {tile}: {us:6.1f}us {2.0 * m * n * k / us / 1e6:5.0f}TF{same}"
        print(line, flush=True)
