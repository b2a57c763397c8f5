"""Sporadic deviation hunt: a FRESH model + runner per repetition (new allocations, cold TLB), NSTEP eager steps from the same
parameters; every workspace buffer after every step is compared with the first repetition's.  Differences at the level of the
embedding-table atomics (<= 1e-5 of the buffer's magnitude) are ignored; the first buffers (creation = execution order) with a
larger deviation name the kernel."""
import gc
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
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
NSTEP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
K.set_default(nt_tile=TILE)
a, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
P = O.synth_params(oarch, seed=21)
batch = O.synth_batch(oarch, B=4, T=8, seed=22, caption_len=32)
ref = {}
for rep in range(REPS + 1):
    m, opt, run = T._runner(a, P)
    pb = m.engine.prepare_batch(batch)
    lab = batch["label"].reshape(-1).to(torch.int32).to("cuda:0")
    m._fresh_shadows(); m._sync_requires_grad()
    for s in range(NSTEP):
        out = run.run(pb, lab, device_step=True)
        torch.cuda.synchronize()
        bufs = {k: v.clone().cpu() for k, v in list(run.eng.buf.items()) + [("head." + k2, v2) for k2, v2 in run.head.buf.items()]}
        bufs["GRAD"] = m.store.grad.clone().cpu()
        bufs["FLAT"] = m.store.flat.clone().cpu()
        if rep == 0:
            ref[s] = bufs
            continue
        bad = []
        for k, v in bufs.items():
            r = ref[s].get(k)
            if r is None or r.shape != v.shape:
                continue
            d = float((v.float() - r.float()).abs().max())
            mag = float(r.float().abs().max())
            if d > 1e-5 * max(mag, 1e-20):
                bad.append((k, tuple(v.shape), int((v.float() != r.float()).sum()), d, mag))
        print(f"rep {rep} step {s + 1}: loss {float(out['loss1']) + float(out['loss2']):.9g}  deviating buffers: {len(bad)}", bad[:8], flush=True)
    del m, opt, run, pb
    gc.collect()
    torch.cuda.empty_cache()
print("done")
