"""Which part of the TRAINING step is not reproducible?  From one snapshot of (parameters, Adam moments) run two eager steps again
and again; after each step compare parameters / moments / gradient bit for bit with the first repetition, tensor by tensor (the
embedding tables, whose gradient scatters are fp32 atomics, are listed separately)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_bench_path_gpu as T  # noqa: E402
from oracle import tvts_oracle as O  # noqa: E402
from tvts_amd import arch as A, hip as K  # noqa: E402

TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
NSTEP = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K.set_default(nt_tile=TILE)
a, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
P = O.synth_params(oarch, seed=21)
batch = O.synth_batch(oarch, B=4, T=8, seed=22, caption_len=32)
m, opt, run = T._runner(a, P)
pb = m.engine.prepare_batch(batch)
lab = batch["label"].reshape(-1).to(torch.int32).to("cuda:0")
m._fresh_shadows(); m._sync_requires_grad()
st = m.store
snap = dict(flat=st.flat.clone(), m=st.m.clone(), v=st.v.clone())
EMB = ("text_positional_embedding", "text_token_embedding.weight", "video_model.class_embedding", "video_model.positional_embedding",
       "video_model.temporal_embedding", "pred_model.type_embed")


def restore():
    st.flat.copy_(snap["flat"]); st.m.copy_(snap["m"]); st.v.copy_(snap["v"])
    opt.step_dev.zero_(); opt.global_step = 0
    st.refresh_shadows()
    torch.cuda.synchronize()


def state():
    return dict(flat=st.flat.clone(), m=st.m.clone(), v=st.v.clone(), grad=st.grad.clone(), shadow=st.shadow.clone().float(),
                shadow_t=st.shadow_t.clone().float())


ref = []
bufref = {}
QUIET = os.environ.get("QUIET", "1") == "1"
for rep in range(REPS + 1):
    restore()
    for s in range(NSTEP):
        out = run.run(pb, lab, device_step=True)
        torch.cuda.synchronize()
        cur = state()
        bufs = {k: v.clone() for k, v in list(run.eng.buf.items()) + [("head." + k2, v2) for k2, v2 in run.head.buf.items()]}
        if rep == 0:
            ref.append(cur)
            bufref[s] = bufs
            continue
        badb = [(k, tuple(v.shape), int((v.float() != bufref[s][k].float()).sum()), float((v.float() - bufref[s][k].float()).abs().max()))
                for k, v in bufs.items() if k in bufref[s] and v.shape == bufref[s][k].shape and not torch.equal(v, bufref[s][k])]
        if badb:
            print(f"rep {rep} step {s + 1}: {len(badb)} workspace buffers differ; first in creation order:", badb[:10], flush=True)
        if QUIET:
            continue
        for key in ("grad", "flat", "m", "v", "shadow"):
            d = cur[key] - ref[s][key]
            if float(d.abs().max()) == 0:
                continue
            emb, other = [], []
            for name in st.shapes:
                o, n = st.off[name], st._n(name)
                dd = d[o:o + n]
                mx = float(dd.abs().max())
                if mx > 0:
                    (emb if name in EMB else other).append((name, int((dd != 0).sum()), mx, float(ref[s][key][o:o + n].abs().max())))
            if other:
                print(f"rep {rep} step {s + 1} {key}: NON-embedding tensors differ:", other[:6], "| embeddings:", [e[0] for e in emb], flush=True)
            elif key != "grad" and any(e[2] > 1e-6 * max(e[3], 1e-30) for e in emb):
                print(f"rep {rep} step {s + 1} {key}: embedding tables", [(e[0], e[1], e[2]) for e in emb], flush=True)
        d = cur["shadow_t"] - ref[s]["shadow_t"]
        if float(d.abs().max()) > 0:
            print(f"rep {rep} step {s + 1} shadow_t differs in {int((d != 0).sum())} elements", flush=True)
print("done")
