"""Which kernel of the step is not bit-reproducible?  Forward + backward of one batch repeated on identical inputs (no optimizer
step); after every repetition each named workspace buffer of the engine and the flat gradient are compared bit for bit with the
first run's.  Buffers are reported in creation (= execution) order, so the first name that differs points at the kernel."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tvts_oracle as O  # noqa: E402  (synthetic parameters / batch only)
from tvts_amd import arch as A, hip as K  # noqa: E402
from tvts_amd.engine import LossHead  # noqa: E402
from tvts_amd.model._common import TVTSv2Base  # noqa: E402

TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K.set_default(nt_tile=TILE)
a, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
P = O.synth_params(oarch, seed=21)
batch = O.synth_batch(oarch, B=BATCH, T=8, seed=22, caption_len=32)
m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
m.load_state_dict(P, strict=True)
m._fresh_shadows(); m._sync_requires_grad()
pb = m.engine.prepare_batch(batch)
head = LossHead(m.store.device)
lab = batch["label"].reshape(-1).to(torch.int32).to("cuda:0")


def bits(t):
    return t.view(torch.int32) if t.dtype == torch.float32 else t.view(torch.int16) if t.dtype == torch.bfloat16 else t


def run():
    m.store.grad.zero_()
    te, ve, pred = m.engine.forward(pb)
    l1, dv, dt = head.contrastive(ve, te)
    l2, dp = head.sorting(pred, lab)
    m.engine.backward(dt, dv, dp)
    torch.cuda.synchronize()


run()
ref = {k: v.clone() for k, v in m.engine.buf.items()}
gref = m.store.grad.clone()
order = list(m.engine.buf.keys())
for i in range(REPS):
    run()
    bad = []
    for k in order:
        v = m.engine.buf[k]
        if not torch.equal(bits(v), bits(ref[k])):
            d = (v.float() - ref[k].float())
            nz = d.nonzero()
            bad.append((k, tuple(v.shape), int(nz.shape[0]), nz[:3].tolist(), float(d.abs().max())))
    gbad = not torch.equal(bits(m.store.grad), bits(gref))
    if bad or gbad:
        print(f"rep {i}: {len(bad)} buffers differ; grad differs: {gbad}")
        for b in bad[:12]:
            print("   ", b)
        if gbad:
            d = (m.store.grad - gref)
            for name in m.store.shapes:
                o, nn = m.store.off[name], m.store._n(name)
                dd = d[o:o + nn]
                if float(dd.abs().max()) > 0:
                    print("    grad", name, int((dd != 0).sum()), float(dd.abs().max()), float(gref[o:o + nn].abs().max()))
        sys.stdout.flush()
print("done", REPS, "repetitions")
