#!/usr/bin/env python3
"""Which parameter gradients differ between two identical full-size engine steps (dev tool, GPU).  python experiments/dbg/full_repro.py [PAIRS=192]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import tvts_oracle as O  # noqa: E402  (synthetic batch / parameters only)
from tvts_amd import arch as A  # noqa: E402
from tvts_amd.engine import LossHead  # noqa: E402
from tvts_amd.model._common import TVTSv2Base  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
ARGS = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
a, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
m = TVTSv2Base(ARGS, arch=a)
m.load_state_dict(O.synth_params(oarch, seed=11), strict=True)
m._fresh_shadows(); m._sync_requires_grad()
head = LossHead(m.store.device)
batch = O.synth_batch(oarch, B=B, T=8, seed=31, caption_len=32)
pb = m.engine.prepare_batch(batch)
lab = batch["label"].reshape(-1).to(torch.int32).to("cuda:0")


def run():
    m.store.grad.zero_()
    te, ve, pred = m.engine.forward(pb)
    loss1, dv, dt = head.contrastive(ve, te)
    loss2, dpred = head.sorting(pred, lab)
    m.engine.backward(dt, dv, dpred)
    torch.cuda.synchronize()
    return {n: m.store.g(n).clone() for n, _ in m.named_parameters()}


g0 = run()
for r in range(3):
    g1 = run()
    bad = [(n, float((g0[n] - g1[n]).abs().max()), float(g0[n].abs().max())) for n in g0 if not torch.equal(g0[n], g1[n])]
    print(f"run {r}: {len(bad)} of {len(g0)} gradients differ")
    for n, d, s in bad[:40]:
        print(f"   {n:60s} max |d| {d:.3e}  (max |g| {s:.3e})")
