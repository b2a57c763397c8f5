import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tvts_amd import hip as K
DEV = "cuda:0"
torch.manual_seed(0)
def run(a, b, bias, res, gh, tile, M, N):
    outs = {}
    def mk(dt): return torch.full((M, N), float("nan"), dtype=dt, device=DEV)
    o = mk(torch.float32); K.gemm_nt(a, b, o, bias=bias, residual=res, tile=tile); outs["f32+res"] = o
    o = mk(torch.bfloat16); K.gemm_nt(a, b, o, tile=tile); outs["plain"] = o
    o = mk(torch.bfloat16); K.gemm_nt(a, b, o, bias=bias, residual=res, tile=tile); outs["bf16+res"] = o
    o, pre = mk(torch.bfloat16), mk(torch.bfloat16); K.gemm_nt(a, b, o, bias=bias, act="quick_gelu", preact=pre, tile=tile); outs["qg"] = o; outs["pre"] = pre
    o = mk(torch.bfloat16); K.gemm_nt(a, b, o, bias=bias, act="gelu", tile=tile); outs["gelu"] = o
    for ga in ("quick_gelu", "gelu", "add"):
        o = mk(torch.bfloat16); K.gemm_nt(a, b, o, gate_h=gh, gate_act=ga, tile=tile); outs["gate_" + ga] = o
    torch.cuda.synchronize()
    return outs
for (M, N, K_) in [(5856, 768, 768), (1537, 520, 64), (100, 132, 128)]:
    a = torch.randn(M, K_, device=DEV).bfloat16(); b = (torch.randn(N, K_, device=DEV) * K_ ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV); res = torch.randn(M, N, device=DEV); gh = torch.randn(M, N, device=DEV).bfloat16()
    ref = run(a, b, bias, res, gh, "128noring", M, N)
    for ring in ("ring2", "ring3", "ring4"):
        got = run(a, b, bias, res, gh, ring, M, N)
        again = run(a, b, bias, res, gh, ring, M, N)
        for k in ref:
            d = (ref[k].float() - got[k].float()).abs()
            nbad = int((ref[k] != got[k]).sum()) if not torch.equal(ref[k], got[k]) else 0
            rep = torch.equal(got[k], again[k])
            if nbad or not rep:
                idx = (ref[k] != got[k]).nonzero()[:4].tolist()
                print(M, N, K_, ring, k, "differs", nbad, "max", float(d.max()), "repeatable", rep, idx, flush=True)
    print(M, N, K_, "done", flush=True)
