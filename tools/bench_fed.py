#!/usr/bin/env python3
"""The step fed from HOST memory every iteration (the trainer's situation; bench.py times inputs that are resident in HBM): eager
launches, prepare_batch (dtype / device normalisation, token sort, H2D copies) inside the timed loop.  fp32 clips as the reference's
loader hands them over (pageable / pinned) and the uint8 wire format of SURVEY 8f N3.  Dev tool, GPU only.  usage: bench_fed.py [PAIRS]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import arch as A  # noqa: E402
from tvts_amd.data_loader import synth_batch  # noqa: E402
from tvts_amd.model._common import TVTSv2Base  # noqa: E402
from tvts_amd.optim import FusedHFAdamW  # noqa: E402
from tvts_amd.step import StepRunner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
a = dict(A.ARCHS["B_16"])
m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a, init_seed=0)
groups = [[], [], [], []]
for name, p in m.named_parameters():
    gi = A.param_group_of(name, a)
    if gi < 0:
        p.requires_grad = False
    else:
        groups[gi].append(p)
opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0], weight_decay=A.GROUP_HPARAMS[i][1]) for i in range(4)], m.store, model=m)
run = StepRunner(m, opt)
m._fresh_shadows(); m._sync_requires_grad()
pool = [synth_batch(a, B, 8, seed=i, caption_len=32) for i in range(2)]


def variant(kind):
    out = []
    for b in pool:
        b = dict(b)
        if kind == "fp32 pinned":
            b["video"] = b["video"].pin_memory()
        elif kind.startswith("uint8"):
            v = (b["video"] * 40 + 128).clamp(0, 255).to(torch.uint8).permute(0, 1, 3, 4, 2).contiguous()  # [B, T, H, W, 3] frames
            b["video"] = v.pin_memory() if "pinned" in kind else v
        out.append(b)
    return out


for kind in ("resident", "fp32 pageable", "fp32 pinned", "uint8 pinned"):
    bs = variant(kind)
    pbs = [m.engine.prepare_batch(b) for b in bs] if kind == "resident" else None
    labs = [b["label"].reshape(-1).to(torch.int32).to("cuda:0") for b in bs]
    for it in range(3):
        run.run(pbs[it % 2] if pbs else m.engine.prepare_batch(bs[it % 2]), labs[it % 2])
    torch.cuda.synchronize()
    n = 10
    t = time.time()
    for it in range(n):
        run.run(pbs[it % 2] if pbs else m.engine.prepare_batch(bs[it % 2]), labs[it % 2])
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(f"{kind:14s}: {dt * 1e3:7.2f} ms per step = {B / dt:7.1f} pairs/s (eager launches, {B} pairs)", flush=True)
