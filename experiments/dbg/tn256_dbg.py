import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import numpy as np
from tvts_amd import _lib, hip as K
lib = _lib.load()
dev = "cuda:0"
np.set_printoptions(linewidth=200)
for (M, N) in ((64, 256), (200, 512)):
    print("==== M", M, "N", N)
    for probe in (0, 1, 16, 63, 70):
        pp = torch.zeros(M, N, device=dev); pp[probe, :] = 1
        qq = torch.zeros(M, N, device=dev); qq[probe, :] = torch.arange(N, device=dev).float() % 64 + 1
        o2 = torch.full((N, N), float("nan"), device=dev)
        K.gemm_tn(pp.bfloat16(), qq.bfloat16(), o2, accumulate=False, tile=256)
        torch.cuda.synchronize()
        exp = (torch.arange(N, device=dev).float() % 64 + 1)[None, :].expand(N, N)
        bad = ~(o2 == exp)
        n16 = (N + 15) // 16
        padded = torch.zeros(n16 * 16, n16 * 16, dtype=torch.bool, device=dev); padded[:N, :N] = bad
        blk = padded.reshape(n16, 16, n16, 16).any(dim=3).any(dim=1)
        print("probe m", probe, "bad elements", int(bad.sum()), "bad 16x16 blocks [a-tile rows x b-tile cols]:")
        if bad.any():
            print(blk.int().cpu().numpy())
            ia, ib = torch.nonzero(bad)[0].tolist()
            print("  first bad (a,b)", ia, ib, "got", o2[ia, ib:ib + 8].tolist())
