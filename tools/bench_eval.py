#!/usr/bin/env python3
"""Throughput of the "next" rows built around the step: the validation forward + on-device retrieval metrics
(N1), the downstream inference model (N2) and the step fed with uint8 frames (N3).  One JSON line each (dev tool, GPU)."""
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import arch as A  # noqa: E402
from tvts_amd.data_loader import synth_batch  # noqa: E402
from tvts_amd.model import metric as M  # noqa: E402
from tvts_amd.model._common import TVTSv2Base, sim_matrix  # noqa: E402


def timed(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    B = int(os.environ.get("PAIRS", "128"))
    # N1: eval forward of the training model (B/16, 8 frames, mask 0.5, 4 captions + sort head) and the metrics
    a = A.ARCHS["B_16"]
    m = TVTSv2Base(args, arch=a)
    m.eval()
    pb = m.engine.prepare_batch(synth_batch(a, B, 8, seed=1))
    m._fresh_shadows()
    with torch.no_grad():
        dt = timed(lambda: m.engine.forward(pb), 5)
    print(json.dumps({"what": "N1 eval forward, ViT-B/16 8-frame mask 0.5, 4 captions + sort head", "pairs_per_gpu": B,
                      "pairs_per_s": B / dt, "ms": 1e3 * dt}))
    g = torch.Generator().manual_seed(0)
    te, ve = torch.randn(4096, 512, generator=g).cuda(), torch.randn(4096, 512, generator=g).cuda()
    dt = timed(lambda: (M.t2v_metrics(sim_matrix(te, ve)), M.v2t_metrics(sim_matrix(te, ve))), 5)
    print(json.dumps({"what": "N1 sim_matrix + t2v/v2t metrics on 4096 x 4096 (device ranks, host summary)", "ms": 1e3 * dt}))
    # N3: the training step fed with uint8 frames vs the fp32 clip
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    groups = [[], [], [], []]
    for name, p in m.named_parameters():
        gi = A.param_group_of(name, a)
        if gi < 0:
            p.requires_grad = False
        else:
            groups[gi].append(p)
    opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0], weight_decay=A.GROUP_HPARAMS[i][1]) for i in range(4)],
                       m.store, model=m)
    runner = StepRunner(m, opt)
    b32 = synth_batch(a, B, 8, seed=2)
    frames = torch.randint(0, 256, (B, 8, 268, 268, 3), dtype=torch.uint8)
    crop = torch.randint(0, 268 - 224 + 1, (B, 2))
    m._sync_requires_grad()
    labels = b32["label"].reshape(-1).to(torch.int32).cuda()
    for name, data in (("fp32 [B,T,3,224,224]", b32), ("uint8 [B,T,268,268,3] + crop", dict(b32, video=frames, crop=crop))):
        pbx = m.engine.prepare_batch(data)
        dt = timed(lambda: runner.run(pbx, labels), 4)
        vb = data["video"].numel() * data["video"].element_size()
        print(json.dumps({"what": "N3 train step, input " + name, "pairs_per_gpu": B, "pairs_per_s": B / dt, "ms": 1e3 * dt,
                          "input_MB": vb / 1e6}))
    del m, runner, opt
    torch.cuda.empty_cache()
    # N2: downstream inference model, 12 unmasked frames, one caption per video
    from tvts_amd.downstream.model_TVTSv2_ViT_B_16 import TVTSv2_B_16
    d = TVTSv2_B_16(pretrained=False)
    Bd = max(8, B // 4)
    db = synth_batch(dict(a, mask_ratio=0.0), Bd, 12, seed=3, n_trans=1)
    pbd = d.engine.prepare_batch(db)
    d._fresh_shadows()
    with torch.no_grad():
        dt = timed(lambda: d.engine.forward(pbd), 4)
    print(json.dumps({"what": "N2 downstream forward, ViT-B/16 12 frames x 196 patches (S = 2353), 1 caption", "pairs_per_gpu": Bd,
                      "pairs_per_s": Bd / dt, "ms": 1e3 * dt}))


if __name__ == "__main__":
    main()
