#!/usr/bin/env python3
"""One weight-gradient GEMM shape, three launches (for tools/pmc.sh counter passes).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import hip as K  # noqa: E402

M, na, nb = (int(x) for x in os.environ.get("SHAPE", "150720,2304,768").split(","))
p = torch.randn(M, na, device="cuda:0").bfloat16()
q = torch.randn(M, nb, device="cuda:0").bfloat16()
out = torch.empty(na, nb, device="cuda:0")
cs = torch.zeros(na, device="cuda:0")
for _ in range(3):
    K.gemm_tn(p, q, out, accumulate=False, colsum=cs)
torch.cuda.synchronize()
