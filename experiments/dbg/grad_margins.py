"""Margins of the parity gates on the headline architecture (B/16, T=8, mask .5, B pairs) against the fp32 CPU oracle: row cosine /
rel-L2 of the embeddings, loss differences, gradient-norm deviation and the worst per-tensor gradient cosine -- for the three
precisions of the residual stream / its gradient (arch["bf16_residual"], arch["bf16_grad_stream"])."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tvts_oracle as O  # noqa: E402  (checker)
from tvts_amd import arch as A  # noqa: E402
from tvts_amd.engine import LossHead  # noqa: E402
from tvts_amd.model._common import TVTSv2Base  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a0, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
P = O.synth_params(oarch, seed=11)
batch = O.synth_batch(oarch, B=B, T=8, seed=12, caption_len=32)
torch.set_num_threads(min(64, torch.get_num_threads()))
leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
r1, r2, rte, rve, rpred = O.step_losses(leaves, batch, oarch)
(r1 + r2).backward()
grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
gn_ref = sum(float(g.double().norm()) ** 2 for g in grads.values()) ** 0.5


def cosmin(x, y):
    x, y = x.detach().double().cpu(), y.detach().double().cpu()
    return float(torch.nn.functional.cosine_similarity(x, y, dim=1).min())


for lowp in ("fp32 residual + bf16 gradient stream (default)", "bf16 residual + bf16 gradient stream (opt-in)", "fp32 streams (rounds 1-2)"):
    a = dict(a0, bf16_residual=lowp.startswith("bf16 residual"), bf16_grad_stream=not lowp.startswith("fp32 streams"))
    m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
    m.load_state_dict(P, strict=True)
    m._fresh_shadows(); m._sync_requires_grad()
    pb = m.engine.prepare_batch(batch)
    m.store.grad.zero_()
    te, ve, pred = m.engine.forward(pb)
    head = LossHead(m.store.device)
    l1, dv, dt = head.contrastive(ve, te)
    l2, dp = head.sorting(pred, batch["label"].reshape(-1).to(torch.int32).to("cuda:0"))
    m.engine.backward(dt, dv, dp)
    torch.cuda.synchronize()
    gn, worst = 0.0, []
    for k, g in grads.items():
        mine = m.store.g(k).detach().cpu()
        gn += float(mine.double().norm()) ** 2
        if float(g.norm()) > 1e-3 * gn_ref:
            worst.append((float(torch.nn.functional.cosine_similarity(mine.double().flatten(), g.double().flatten(), dim=0)), k))
    worst.sort()
    print(f"{lowp}: B={B} te rel-L2 {float((te.cpu().double() - rte.double()).norm() / rte.double().norm()):.4f} ve rel-L2 {float((ve.cpu().double() - rve.double()).norm() / rve.double().norm()):.4f} te cos {cosmin(te, rte):.6f} ve cos {cosmin(ve, rve):.6f} d loss1 {float(l1) - float(r1):+.2e} "
          f"d loss2 {float(l2) - float(r2):+.2e} grad-norm {gn ** 0.5:.5f} vs {gn_ref:.5f} ({(gn ** 0.5 / gn_ref - 1) * 100:+.3f} %) "
          f"worst tensor cosines {[(round(c, 5), k) for c, k in worst[:4]]}", flush=True)
    del m
