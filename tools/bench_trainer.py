#!/usr/bin/env python3
"""One epoch of the reference-shaped TRAINER (tvts_amd.trainer.Trainer_TVTSv2_B_16._train_epoch: tokenised batches from a loader in
host memory, prepare_batch + step per iteration, the per-epoch loss log) on ViT-B/16, 8 frames: the rate a user of the drop-in sees,
beside bench.py's resident-input step.  Two timed epochs per setting, the faster one reported (an epoch's first step cannot hide
its own host-to-device copy: with STEPS steps per epoch that is 1 / STEPS of the clip copy per step, so use >= 40 steps).
TVTS_TRAINER_GRAPH=0: every step eager.  Dev tool, GPU only.  usage: bench_trainer.py [PAIRS=192] [STEPS=40]"""
import logging
import os
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tvts_amd import arch as A  # noqa: E402
from tvts_amd.data_loader import synth_batch  # noqa: E402
from tvts_amd.model import metric as M  # noqa: E402
from tvts_amd.model._common import TVTSv2Base  # noqa: E402
from tvts_amd.model.loss import NormSoftmaxLoss  # noqa: E402
from tvts_amd.optim import FusedHFAdamW  # noqa: E402
from tvts_amd.trainer.trainer import Trainer_TVTSv2_B_16  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40


class Loader(list):
    def __init__(self, batches, name, batch_size):
        super().__init__(batches)
        self.dataset_name, self.batch_size, self.n_samples = name, batch_size, batch_size * len(batches)


class Config(dict):
    resume = None

    def __init__(self, save_dir):
        super().__init__(trainer=dict(epochs=1, save_period=10, verbosity=2, monitor="off", init_val=False),
                         arch=dict(type="TVTSv2_B_16", args={}), optimizer=dict(type="AdamW", args=dict(lr=1e-4)))
        self.save_dir = save_dir

    def get_logger(self, name, verbosity=2):
        return logging.getLogger(name)


a = A.ARCHS["B_16"]
args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1, schedule=[])
m = TVTSv2Base(args, arch=a, init_seed=0)
groups = [[], [], [], []]
for name, p in m.named_parameters():
    gi = A.param_group_of(name, a)
    if gi < 0:
        p.requires_grad = False
    else:
        groups[gi].append(p)
opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0], weight_decay=A.GROUP_HPARAMS[i][1]) for i in range(4)], m.store, model=m)
pool = [synth_batch(a, B, 8, seed=i, caption_len=32) for i in range(2)]
for pinned in (False, True):
    if pinned:
        for b in pool:
            b["video"] = b["video"].pin_memory()
    yt = Loader([dict(pool[i % 2]) for i in range(STEPS)], "YTTemporal", B)
    tr = Trainer_TVTSv2_B_16(args, m, NormSoftmaxLoss(), [M.t2v_metrics, M.v2t_metrics], opt, config=Config(tempfile.mkdtemp()),
                             data_loader=[yt], valid_data_loader=None, max_samples_per_epoch=10 ** 9)
    m.train()
    tr._train_epoch(0)  # warm-up epoch (workspaces; graph capture of the batch signature)
    torch.cuda.synchronize()
    dts = []
    for ep in (1, 2):
        t = time.time()
        log = tr._train_epoch(ep)
        torch.cuda.synchronize()
        dts.append((time.time() - t) / STEPS)
    dt = min(dts)
    rp = tr.replay
    print(f"trainer epoch, {B} pairs x {STEPS} steps, fp32 clips from {'pinned' if pinned else 'pageable'} host memory: {dt * 1e3:7.2f} ms per step = "
          f"{B / dt:7.1f} pairs/s  (epochs {', '.join(f'{x * 1e3:.2f}' for x in dts)} ms; replays {rp.replays}, eager {rp.eager}; epoch loss {log['loss_0']:.4f})", flush=True)
