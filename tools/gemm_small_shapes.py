import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tvts_amd import hip as K, _lib
dev = "cuda:0"
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
shapes = [(6144,1024,1024),(6144,3072,1024),(6144,4096,1024),(6144,1024,4096),(6144,1024,3072),(24576,512,512),(24576,1536,512),(24576,2048,512),(24576,512,2048)]
for (m,n,k) in shapes:
    a = torch.randn(m,k,device=dev).bfloat16(); b=(torch.randn(n,k,device=dev)*k**-0.5).bfloat16(); out=torch.empty(m,n,dtype=torch.bfloat16,device=dev)
    # rotate buffers to defeat cache residency
    As=[a.clone() for _ in range(4)]; Os=[out.clone() for _ in range(4)]
    line=f"{m}x{n}x{k}:"
    for tile in (128,256):
        i=[0]
        def f():
            i[0]=(i[0]+1)%4
            K.gemm_nt(As[i[0]], b, Os[i[0]], tile=tile)
        ms=timeit(f)
        line+=f"  tile{tile} {ms*1e3:6.1f}us {2.0*m*n*k/ms/1e9:6.0f}TF"
    print(line, flush=True)
