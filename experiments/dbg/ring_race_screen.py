"""Race screen of the ring kernels: many launches per shape (back to back, different buffers, a second stream hammering HBM beside them),
every result compared bit for bit with the double-buffered 128 kernel's."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tvts_amd import hip as K
dev = "cuda:0"
torch.manual_seed(1)
bad = 0
noise = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
shapes = [(1536, 512, 2048), (1536, 1536, 512), (1570, 768, 768), (1570, 2304, 768), (3072, 2048, 512), (5856, 768, 3072), (977, 520, 192), (6144, 1024, 4096), (300, 132, 64)]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device=dev).bfloat16(); b = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev); res = torch.randn(m, n, device=dev)
    ref = torch.empty(m, n, dtype=torch.float32, device=dev)
    K.gemm_nt(a, b, ref, bias=bias, residual=res, tile="128noring")
    refb = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    K.gemm_nt(a, b, refb, bias=bias, act="quick_gelu", tile="128noring")
    torch.cuda.synchronize()
    outs = [torch.empty_like(ref) for _ in range(8)]; outsb = [torch.empty_like(refb) for _ in range(8)]
    nbad = 0
    for it in range(40):
        with torch.cuda.stream(side):
            noise.add_(1)  # HBM traffic beside the GEMMs
        for tile in ("ring2", "ring3", "ring"):
            for j in range(8):
                K.gemm_nt(a, b, outs[j], bias=bias, residual=res, tile=tile)
                K.gemm_nt(a, b, outsb[j], bias=bias, act="quick_gelu", tile=tile)
            torch.cuda.synchronize()
            for j in range(8):
                if not torch.equal(outs[j], ref) or not torch.equal(outsb[j], refb):
                    nbad += 1
    print(m, n, k, "launches", 40 * 3 * 16, "mismatching results", nbad, flush=True)
    bad += nbad
print("TOTAL mismatches", bad)
