"""One TVTSv2 pretrain step (v2/trainer/trainer.py:463-512) on the HIP engine, without autograd.

zero_grad -> model forward -> all-gather of embeddings -> sim_matrix + InfoNCE (+ 2*CE sorting loss) ->
hand-written backward (gradients all-reduced range by range while it runs) -> fused HF-AdamW.
"""
from __future__ import annotations

import os
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
        # OPT-IN (arch["adamw_ranges"] / TVTS_ADAMW_RANGES=1): the fused optimizer's update range by range beside the backward
        # (FusedHFAdamW.step_range), each range as soon as its gradient is final.  Built for the reference's per-GPU batches, where
        # the single AdamW launch + the transposes are 1.0 of 15.6 ms, and MEASURED SLOWER at every batch (profiles/r05_adamw_ranges.txt:
        # 12 / 24 / 192 pairs 757 / 987 / 1354 pairs/s with it against 774 / 1022 / 1363 without): an HBM-bound 4.7 GB pass takes the
        # bandwidth and the CUs it runs on away from the backward it was meant to hide under, and its ~18 fork edges cost the replayed
        # graph what the r04 side-stream experiments already showed.  Same bits either way (tests/test_bench_path_gpu.py).
        self.ranged = self.fused and hasattr(optimizer, "step_range") and bool(self.eng.arch.get("adamw_ranges", os.environ.get("TVTS_ADAMW_RANGES", "0") == "1"))
        # exchange diagnostics (bench.py at world > 1, eager launches only): event pairs on the compute stream around the two places where
        # it waits for a collective -- {"gather": [...], "allreduce": [...]} when switched on, None otherwise
        self.diag = None

    def _bracket(self, key, fn):
        """fn() between two events of the current stream when the diagnostics are on: the elapsed time is what the compute stream
        spent WAITING there (the exposed part of the collective), since fn only enqueues waits and small copies"""
        if self.diag is None or torch.cuda.is_current_stream_capturing():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.diag.setdefault(key, []).append((e0, e1))
        return out

    def losses_and_grads(self, pb, te, ve, pred, labels):
        B = pb["B"]
        vg, tg = self._bracket("gather", self.gather.result)  # started by the engine before the sort head ran
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
        if self.ranged:
            self.opt.grad_scale = 1.0 / self.sync.W
            self.opt.begin_ranges(device_step=device_step)
            self.eng.param_ready = self.opt.step_range
        try:
            self.eng.backward(d_te, d_ve, dpred)
        finally:
            self.eng.param_ready = None
        if hasattr(self.eng, "end_step"):
            self.eng.end_step()  # (e4m3 weight gradients: this step's amax values become the next step's per-tensor scales)
        scale = self._bracket("allreduce", self.sync.finish)
        if self.fused:
            self.opt.grad_scale = scale
            self.opt.step(device_step=device_step)
        else:
            if scale != 1.0:
                self.store.grad.mul_(scale)
            self.model._install_grads()
            self.opt.step()
        return dict(loss1=loss1, loss2=loss2)
