#!/usr/bin/env python3
"""dev: the fused AdamW kernel through its host-step and its device-step (+ device hyper table) paths on identical inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tvts_amd import hip as K
n = 4096 * 4
g = torch.Generator().manual_seed(0)
p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
groups = torch.tensor([0] * 4 + [1] * 4 + [2] * 4 + [3] * 4, dtype=torch.uint8).cuda()
lr4, wd4 = [1e-4, 1e-4, 1e-7, 1e-7], [0.05, 0.0, 0.05, 0.0]
res = {}
for mode in ("host", "dev", "dev_hyper"):
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    sh = torch.zeros(n, dtype=torch.bfloat16).cuda()
    sd = torch.zeros(1, dtype=torch.int32).cuda()
    hy = torch.tensor(lr4 + wd4, dtype=torch.float32).cuda()
    for step in (1, 2, 3):
        if mode == "host":
            K.adamw_hf(p, gr.cuda() * step, m, v, sh, groups, lr4, wd4, step)
        else:
            sd.add_(1)
            K.adamw_hf(p, gr.cuda() * step, m, v, sh, groups, lr4, wd4, step, step_dev=sd, hyper_dev=hy if mode == "dev_hyper" else None)
    res[mode] = (p.clone(), m.clone(), v.clone())
for k in ("dev", "dev_hyper"):
    print(k, "vs host: dp %.3e (of %.3e moved) dm %.3e dv %.3e" % (
        float((res[k][0] - res["host"][0]).abs().max()), float((res["host"][0] - p0.cuda()).abs().max()),
        float((res[k][1] - res["host"][1]).abs().max()), float((res[k][2] - res["host"][2]).abs().max())))
    for gi in range(4):
        sl = slice(gi * 4096, (gi + 1) * 4096)
        a, b = (res[k][0][sl] - p0.cuda()[sl]), (res["host"][0][sl] - p0.cuda()[sl])
        print("   group", gi, "mean |delta| dev %.4e host %.4e ratio %.6f" % (float(a.abs().mean()), float(b.abs().mean()), float(a.abs().mean() / b.abs().mean())))
