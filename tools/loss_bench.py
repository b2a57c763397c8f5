#!/usr/bin/env python3
"""Timing of the contrastive head at the gathered batch sizes of 1..8 GPUs x 192 pairs (and 4096) (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvts_amd.engine import LossHead

def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for G in (192, 768, 1536, 4096):
    for E in (512,):
        v, t = torch.randn(G, E, device="cuda"), torch.randn(G, E, device="cuda")
        head = LossHead(torch.device("cuda"))
        print(f"G={G:5d} E={E}: contrastive fwd+bwd {timeit(lambda: head.contrastive(v, t)):8.1f} us")
