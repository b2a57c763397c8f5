#!/usr/bin/env python3
"""dev: per-step losses of eager (host step), eager (device step) and hipGraph-replayed steps from one start state."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import tvts_oracle as O
from tvts_amd import arch as A
from tvts_amd.model._common import TVTSv2Base
from tvts_amd.optim import FusedHFAdamW
from tvts_amd.step import StepRunner
from tvts_amd import _lib

ARGS = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
name = sys.argv[1] if len(sys.argv) > 1 else "B_16"
force = int(sys.argv[2]) if len(sys.argv) > 2 else 256
a = A.ARCHS[name] if name in A.ARCHS else A.small_arch()
oarch = O.ARCHS[name] if name in A.ARCHS else O.tiny_arch(**a)
P = O.synth_params(oarch, seed=21)
batch = O.synth_batch(oarch, B=4, T=8 if name in A.ARCHS else 2, seed=22, caption_len=32 if name in A.ARCHS else 9)
from tvts_amd import hip as _K
_K.set_default(nt_tile=force)


def runner():
    m = TVTSv2Base(ARGS, arch=a)
    m.load_state_dict(P, strict=True)
    groups = [[], [], [], []]
    for n, p in m.named_parameters():
        gi = A.param_group_of(n, a)
        if gi < 0: p.requires_grad = False
        else: groups[gi].append(p)
    opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0], weight_decay=A.GROUP_HPARAMS[i][1]) for i in range(4)], m.store, model=m)
    r = StepRunner(m, opt)
    m._fresh_shadows(); m._sync_requires_grad()
    return m, opt, r


lab = batch["label"].reshape(-1).to(torch.int32).cuda()
res = {}
GR = {}
for mode in os.environ.get("MODES", "eager_dev,graph").split(","):
    m, opt, r = runner()
    pb = m.engine.prepare_batch(batch)
    losses, grads = [], []
    if mode == "graph":
        opt.sync_hyper()
        snap = [t.clone() for t in (m.store.flat, m.store.m, m.store.v)]
        r.run(pb, lab, device_step=True); torch.cuda.synchronize()
        m.store.flat.copy_(snap[0]); m.store.m.copy_(snap[1]); m.store.v.copy_(snap[2])
        opt.step_dev.zero_(); opt.global_step = 0
        m.store.refresh_shadows(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = r.run(pb, lab, device_step=True)
        for _ in range(3):
            g.replay(); torch.cuda.synchronize()
            losses.append(float(out["loss1"]) + float(out["loss2"])); grads.append(float(m.store.grad.double().norm()))
            GR.setdefault(mode, []).append({n: m.store.g(n).clone() for n in m.store.shapes})
    else:
        for _ in range(3):
            out = r.run(pb, lab, device_step=(mode == "eager_dev")); torch.cuda.synchronize()
            losses.append(float(out["loss1"]) + float(out["loss2"])); grads.append(float(m.store.grad.double().norm()))
            GR.setdefault(mode, []).append({n: m.store.g(n).clone() for n in m.store.shapes})
    res[mode] = (losses, grads, m.store.flat.clone(), m.store.m.clone())
    print(mode, ["%.6f" % l for l in losses], ["%.6f" % g for g in grads], flush=True)
for st in range(3 if "graph" in GR and "eager_dev" in GR else 0):
    bad = []
    for n in GR["graph"][st]:
        a, b = GR["graph"][st][n].double(), GR["eager_dev"][st][n].double()
        d = float((a - b).norm() / (b.norm() + 1e-12))
        if d > 1e-2 or not torch.isfinite(a).all():
            bad.append((n, d, float(a.abs().max()), float(b.abs().max())))
    print("step", st, "tensors whose graph gradient differs from eager:", len(bad))
    for x in bad[:40]:
        print("   ", x)
base = res[list(res)[0]]
for k, v in res.items():
    print(k, "param maxdiff vs eager_dev %.3e" % float((v[2] - base[2]).abs().max()), "m maxdiff %.3e" % float((v[3] - base[3]).abs().max()))
Pr = {k: v.clone() for k, v in P.items()}
st, curve = {}, []
for _ in range(3):
    r1, r2, _ = O.train_step(Pr, batch, oarch, st)
    curve.append(r1 + r2)
print("oracle", ["%.6f" % c for c in curve])
