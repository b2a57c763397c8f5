#!/usr/bin/env python3
"""dev: ONE optimizer step from the same state through the host-step and the device-step path of FusedHFAdamW: which tensors differ."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import tvts_oracle as O
from tvts_amd import arch as A
from tvts_amd.model._common import TVTSv2Base
from tvts_amd.optim import FusedHFAdamW
from tvts_amd.step import StepRunner
ARGS = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
a = A.small_arch(); oarch = O.tiny_arch(**a)
P = O.synth_params(oarch, seed=21)
batch = O.synth_batch(oarch, B=4, T=2, seed=22, caption_len=9)
lab = batch["label"].reshape(-1).to(torch.int32).cuda()
out = {}
for mode in (False, True, False):
    m = TVTSv2Base(ARGS, arch=a); m.load_state_dict(P, strict=True)
    groups = [[], [], [], []]
    for n, p in m.named_parameters():
        gi = A.param_group_of(n, a)
        if gi < 0: p.requires_grad = False
        else: groups[gi].append(p)
    opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0], weight_decay=A.GROUP_HPARAMS[i][1]) for i in range(4)], m.store, model=m)
    r = StepRunner(m, opt); m._fresh_shadows(); m._sync_requires_grad()
    pb = m.engine.prepare_batch(batch)
    r.run(pb, lab, device_step=mode); torch.cuda.synchronize()
    key = "dev" if mode else ("host" if "host" not in out else "host2")
    out[key] = dict(p={n: m.store.p(n).clone() for n in m.store.shapes}, g={n: m.store.g(n).clone() for n in m.store.shapes},
                    sh=m.store.shadow.clone(), sht=m.store.shadow_t.clone(), mm=m.store.m.clone(), vv=m.store.v.clone(), step=int(opt.step_dev.item()))
    print(key, "step_dev", out[key]["step"], "hyper", opt.hyper_dev.tolist())
for k in ("host2", "dev"):
    print("==", k, "vs host: m maxdiff %.3e v maxdiff %.3e shadow diffs %d shadow_t diffs %d" % (
        float((out[k]["mm"] - out["host"]["mm"]).abs().max()), float((out[k]["vv"] - out["host"]["vv"]).abs().max()),
        int((out[k]["sh"] != out["host"]["sh"]).sum()), int((out[k]["sht"] != out["host"]["sht"]).sum())))
    worst = []
    for n in out["host"]["p"]:
        dg = float((out[k]["g"][n] - out["host"]["g"][n]).abs().max())
        d = out[k]["p"][n] - out["host"]["p"][n]
        mv = (out["host"]["p"][n] - P[n].cuda()).abs().max()
        worst.append((float(d.abs().max()), float(mv), dg, n))
    worst.sort(reverse=True)
    for w in worst[:8]:
        print("   dp %.3e (moved %.3e) dgrad %.3e  %s" % w)
