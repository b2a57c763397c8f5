import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tvts_amd import hip as K
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for M in (4096, 8192, 16384, 50240, 100480):
    p = torch.randn(M, 2304, device="cuda").bfloat16(); q = torch.randn(M, 768, device="cuda").bfloat16()
    out = torch.zeros(2304, 768, device="cuda")
    ms = timeit(lambda: K.gemm_tn(p, q, out, accumulate=True))
    print(f"M={M}: {ms*1e3:.1f} us  per 1K rows {ms*1e3/(M/1024):.2f} us  {2.0*M*2304*768/ms/1e9:.0f} TF")
