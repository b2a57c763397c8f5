#!/usr/bin/env python3
"""Does a sample's forward depend on its position in the batch?  (dev tool, GPU)  python experiments/dbg/perm_repro.py [PAIRS=192]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import tvts_oracle as O  # noqa: E402  (synthetic batch / parameters only)
from tvts_amd import arch as A  # noqa: E402
from tvts_amd.model._common import TVTSv2Base  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
ARGS = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
a, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
NT = oarch["n_trans"]
m = TVTSv2Base(ARGS, arch=a)
m.load_state_dict(O.synth_params(oarch, seed=11), strict=True)
m._fresh_shadows(); m._sync_requires_grad()
batch = O.synth_batch(oarch, B=B, T=8, seed=31, caption_len=32)
perm = torch.roll(torch.arange(B), 37)
bp = {"video": batch["video"][perm], "keep_ind": batch["keep_ind"][perm], "label": batch["label"][perm],
      "text": batch["text"].reshape(NT, B, -1)[:, perm].reshape(NT * B, -1)}
eng = m.engine


def text_trace(bt):
    pb = eng.prepare_batch(bt)
    t = eng.text_forward(pb["ids"], pb["eot_rows"], pb["N"], pb["L"], eot_index=pb.get("eot_index")).clone()
    bufs = {k: v.clone() for k, v in eng.buf.items() if (k.startswith("txt0") or k.startswith("txt.x")) and v.dtype in (torch.float32, torch.bfloat16)}
    return pb, t, bufs


pb0, t0, b0 = text_trace(batch)
pb1, t1, b1 = text_trace(bp)
N, L = pb0["N"], pb0["L"]
cap = (torch.arange(NT)[:, None] * B + perm[None, :]).reshape(-1).to("cuda:0")   # caption row of the rotated batch -> original row
d = (t1.double() - t0[cap].double()).norm(dim=1) / t0[cap].double().norm(dim=1)
print("caption embeddings: rows that differ", int((d > 0).sum()), "of", N, "max rel", float(d.max()))
for k in sorted(b0):
    x0, x1 = b0[k], b1[k]
    if x0.shape != x1.shape or x0.dim() != 2:
        continue
    if x0.shape[0] == N * L:
        rows = (cap[:, None] * L + torch.arange(L, device="cuda:0")[None, :]).reshape(-1)
    elif x0.shape[0] == N:
        rows = cap
    else:
        continue
    dd = (x1.double() - x0[rows].double()).abs().amax(dim=1)
    print(f"  {k:28s} {tuple(x0.shape)} {str(x0.dtype):15s} rows differing {int((dd > 0).sum()):7d}  max |d| {float(dd.max()):.3e}")
x0, x1 = b0["txt0.mid"], b1["txt0.mid"]
rows = (cap[:, None] * L + torch.arange(L, device="cuda:0")[None, :]).reshape(-1)     # new row i holds original row rows[i]
dd = (x1 != x0[rows])
bad = dd.any(dim=1)
new_r = torch.arange(N * L, device="cuda:0")
for name, r in (("new position", new_r), ("old position", rows)):
    h = torch.bincount(((r[bad] % 256) // 16).cpu(), minlength=16)
    print(name, "16-row tile inside the 256-row tile of the differing rows:", h.tolist())
cols = dd.sum(dim=0)
print("differing elements per column block of 16:", cols.reshape(-1, 16).sum(1).tolist())
print("fraction of elements differing in bad rows", float(dd[bad].float().mean()))
