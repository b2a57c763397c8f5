#!/usr/bin/env python3
"""Round 5: which gradients differ between repeated full-size steps / between eager and graph-replayed steps, per engine option
(text tower on its own stream, hybrid stream).  python experiments/dbg/r05_repro.py [PAIRS=192]"""
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
oarch = O.ARCHS["B_16"]
P = O.synth_params(oarch, seed=11)
batch = O.synth_batch(oarch, B=B, T=8, seed=31, caption_len=32)
for text_side, hybrid in ((False, False), (True, False), (False, True), (True, True)):
    a = dict(A.ARCHS["B_16"], text_side=text_side, hybrid_stream=hybrid)
    m = TVTSv2Base(ARGS, arch=a)
    m.load_state_dict(P, strict=True)
    m._fresh_shadows(); m._sync_requires_grad()
    head = LossHead(m.store.device)
    pb = m.engine.prepare_batch(batch)
    lab = batch["label"].reshape(-1).to(torch.int32).to("cuda:0")

    def run():
        m.store.grad.zero_()
        te, ve, pred = m.engine.forward(pb)
        loss1, dv, dt = head.contrastive(ve, te)
        loss2, dpred = head.sorting(pred, lab)
        m.engine.backward(dt, dv, dpred)

    def grads():
        torch.cuda.synchronize()
        return {n: m.store.g(n).clone() for n, _ in m.named_parameters()}
    run(); g0 = grads()
    for r in range(2):
        run(); g1 = grads()
        bad = [(n, float((g0[n] - g1[n]).abs().max()), float(g0[n].abs().max())) for n in g0 if not torch.equal(g0[n], g1[n])]
        print(f"text_side {text_side} hybrid {hybrid} eager run {r}: {len(bad)} of {len(g0)} gradients differ", flush=True)
        for n, d, s in bad[:12]:
            print(f"   {n:60s} max |d| {d:.3e}  (max |g| {s:.3e})")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for r in range(2):
        g.replay(); g1 = grads()
        bad = [(n, float((g0[n] - g1[n]).abs().max()), float(g0[n].abs().max())) for n in g0 if not torch.equal(g0[n], g1[n])]
        print(f"text_side {text_side} hybrid {hybrid} replay {r}: {len(bad)} of {len(g0)} gradients differ", flush=True)
        for n, d, s in bad[:12]:
            print(f"   {n:60s} max |d| {d:.3e}  (max |g| {s:.3e})")
    del g, m, head
    torch.cuda.empty_cache()
