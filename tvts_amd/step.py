"""One TVTSv2 pretrain step (v2/trainer/trainer.py:463-512) on the HIP engine, without autograd.

zero_grad -> model forward -> all-gather of embeddings -> sim_matrix + InfoNCE (+ 2*CE sorting loss) ->
hand-written backward (gradients all-reduced range by range while it runs) -> fused HF-AdamW.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import dist as D
from .engine import LossHead


class StepRunner:
    def __init__(self, model, optimizer, loss_head: Optional[LossHead] = None):
        self.model, self.opt = model, optimizer
        self.eng, self.store = model.engine, model.store
        self.head = loss_head or LossHead(self.store.device)
        self.sync = D.GradSync(self.store.grad)
        self.fused = hasattr(optimizer, "chunk_group")
        self.eng.grad_ready = self.sync.reduce_range if (self.sync.W > 1 or self.sync.native) else None
        self.gather = D.EmbedGather()

    def losses_and_grads(self, pb, te, ve, pred, labels):
        B = pb["B"]
        vg, tg = self.gather.result()  # started by the engine before the sort head ran
        loss1, dv_all, dt_all = self.head.contrastive(vg, tg)
        if pred is not None:
            loss2, dpred = self.head.sorting(pred, labels)
        else:
            loss2, dpred = None, None
        return loss1, loss2, D.local_rows(dt_all, B), D.local_rows(dv_all, B), dpred

    def step(self, data: dict, device_step: bool = False, pb=None):
        m = self.model
        if hasattr(self.eng, "training"):  # v1: DistilBERT's dropout follows the module's train() / eval() flag
            self.eng.training = bool(m.training)
        m._fresh_shadows()
        m._sync_requires_grad()
        if pb is None:
            pb = self.eng.prepare_batch(data)
        labels = data["label"].reshape(-1).to(torch.int32).to(self.store.device) if ("label" in data and pb["NT"] != 1) else None
        return self.run(pb, labels, device_step)

    def run(self, pb, labels, device_step=False):
        """The device-side part of the step (capturable in a hipGraph when world == 1)."""
        if self.sync.W > 1:
            D._apply_cu_reservation(pb["B"] * pb["S"])  # (decided once, from the first batch's token rows per GPU)
        self.store.grad.zero_()
        self.sync.bytes_sent = 0
        self.eng.embeds_ready = self.gather.start  # only the training step gathers; eval / autograd forwards do not
        try:
            te, ve, pred = self.eng.forward(pb)
        finally:
            self.eng.embeds_ready = None
        loss1, loss2, d_te, d_ve, dpred = self.losses_and_grads(pb, te, ve, pred, labels)
        self.eng.backward(d_te, d_ve, dpred)
        if hasattr(self.eng, "end_step"):
            self.eng.end_step()  # (e4m3 weight gradients: this step's amax values become the next step's per-tensor scales)
        scale = self.sync.finish()
        if self.fused:
            self.opt.grad_scale = scale
            self.opt.step(device_step=device_step)
        else:
            if scale != 1.0:
                self.store.grad.mul_(scale)
            self.model._install_grads()
            self.opt.step()
        return dict(loss1=loss1, loss2=loss2)
