"""One TVTSv2 pretrain step (v2/trainer/trainer.py:463-512) on the HIP engine, without autograd.

zero_grad -> model forward -> all-gather of embeddings -> sim_matrix + InfoNCE (+ 2*CE sorting loss) ->
hand-written backward (gradients all-reduced range by range while it runs) -> fused HF-AdamW.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import dist as D
from . import hip as K
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

    MAX_STEPS_IN_FLIGHT = 2

    def _throttle(self):
        """The host may run at most MAX_STEPS_IN_FLIGHT steps ahead of the device.  With page-locked input batches nothing else holds
        it back: every batch it prepares ahead is a fresh device copy of the clip tensor (925 MB of fp32 at 192 pairs) waiting for its
        step, and the allocator grows -- and synchronises -- to hold them (measured: the 192-pair trainer epoch 4 % slower from pinned
        than from pageable memory).  Called at the top of a step; the step's own event is recorded by _stepped()."""
        q = self.__dict__.setdefault("_inflight", [])
        while len(q) >= self.MAX_STEPS_IN_FLIGHT:
            q.pop(0).synchronize()

    def _stepped(self):
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            self.__dict__.setdefault("_inflight", []).append(ev)

    def step(self, data: dict, device_step: bool = False, pb=None):
        out = self._step(data, device_step, pb)
        self._stepped()
        return out

    def _step(self, data: dict, device_step: bool = False, pb=None):
        self._throttle()
        m = self.model
        if hasattr(self.eng, "training"):  # v1: DistilBERT's dropout follows the module's train() / eval() flag
            self.eng.training = bool(m.training)
        m._fresh_shadows()
        m._sync_requires_grad()
        if pb is None:
            pb = self.eng.prepare_batch(data)
        return self.run(pb, self._labels(data, pb), device_step)

    def _labels(self, data, pb):
        """int32 device labels of the sorting loss: prepare_batch stages them with the batch's other index tensors (v2 engine)"""
        if pb["NT"] == 1 or "label" not in data:
            return None
        if pb.get("labels") is not None:
            return pb["labels"]
        return data["label"].reshape(-1).to(torch.int32).to(self.store.device)

    def run(self, pb, labels, device_step=False):
        """The device-side part of the step (capturable in a hipGraph when world == 1)."""
        if self.sync.W > 1:
            D._apply_cu_reservation(pb["B"] * pb["S"])  # (decided once, from the first batch's token rows per GPU)
        K.zero_(self.store.grad)
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


def _map_tensors(x, fn):
    """fn over every tensor of a (nested) batch structure; everything else is kept"""
    if torch.is_tensor(x):
        return fn(x)
    if isinstance(x, dict):
        return {k: _map_tensors(v, fn) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return type(x)(_map_tensors(v, fn) for v in x)
    return x


def _signature(x):
    if torch.is_tensor(x):
        return ("t", tuple(x.shape), str(x.dtype))
    if isinstance(x, dict):
        return tuple((k, _signature(v)) for k, v in sorted(x.items()))
    if isinstance(x, (tuple, list)):
        return tuple(_signature(v) for v in x)
    return x if isinstance(x, (int, float, bool, str, type(None))) else type(x).__name__


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (tuple, list)):
        for d, s_ in zip(dst, src):
            _copy_into(d, s_)


class GraphReplay:
    """The trainer loop's step as a replayed hipGraph (what bench.py does with its two resident batches, for batches that ARRIVE):
    one graph per batch signature -- the shapes and types of everything prepare_batch hands the device step; the reference's loop
    alternates a YT-Temporal and a WebVid loader (v2/trainer/trainer.py:463), i.e. two signatures, a last partial batch makes a third.

    A signature's first step runs eagerly (its workspaces are allocated there), its second is captured over device-side COPIES of the
    batch's tensors, and from then on a step is: prepare_batch (host-to-device on the engine's copy stream, under the previous step)
    -> device-to-device copy into the captured buffers (on the compute stream, i.e. behind the previous replay: 0.3 ms for a
    192-pair fp32 clip tensor) -> replay.  The learning rate and weight decay travel through FusedHFAdamW.sync_hyper (read by the
    captured AdamW launch from device memory), the step counter lives on the device.  Same launches, same order, same bits as the
    eager step (tests/test_trainer_gpu.py::test_graph_replay_in_the_trainer_loop_is_the_eager_loop).

    World 1 with the fused optimizer only (a captured multi-rank step has never run on hardware: bench.py keeps it opt-in too);
    anything else, and any failure to capture, falls back to StepRunner.step for good.  OPT-IN (TVTS_TRAINER_GRAPH=1): with
    prepare_batch free of synchronising copies the eager loop already keeps the device busy back to back, and the replayed loop
    measures 1 % (192 pairs) to 3 - 4 % (12 pairs) SLOWER than it (profiles/r06_bench_product_path*.txt)."""

    MAX_SIGNATURES = 6

    def __init__(self, runner: StepRunner):
        self.r = runner
        self.cache = {}
        self.usable = (torch.cuda.is_available() and runner.fused and not runner.ranged and runner.sync.W == 1 and not runner.sync.native
                       and os.environ.get("TVTS_TRAINER_GRAPH", "0") == "1")
        self.replays = self.captures = self.eager = 0

    def step(self, data: dict):
        r = self.r
        if not self.usable:
            self.eager += 1
            return r.step(data)
        out = self._step(data)
        r._stepped()
        return out

    def _step(self, data: dict):
        r = self.r
        r._throttle()
        m = r.model
        if hasattr(r.eng, "training"):
            r.eng.training = bool(m.training)
        m._fresh_shadows()
        m._sync_requires_grad()
        pb = r.eng.prepare_batch(data)
        labels = r._labels(data, pb)
        sig = (_signature(pb), None if labels is None else tuple(labels.shape), bool(m.training),
               hash(tuple(r.eng.requires_grad.values())) if hasattr(r.eng, "requires_grad") else None)
        ent = self.cache.get(sig)
        if ent is None:
            if len(self.cache) >= self.MAX_SIGNATURES:
                self.cache.pop(next(iter(self.cache)))
            self.cache[sig] = dict(graph=None)
            self.eager += 1
            return r.run(pb, labels, device_step=True)   # first sight: eager (allocates this signature's workspaces)
        self.cache[sig] = self.cache.pop(sig)            # most recently used last
        if ent["graph"] is None:
            try:
                ent["pb"] = _map_tensors(pb, lambda t: t.clone())
                ent["labels"] = None if labels is None else labels.clone()
                r.opt.sync_hyper()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    ent["out"] = r.run(ent["pb"], ent["labels"], device_step=True)
                ent["graph"] = g
                self.captures += 1
            except Exception as e:  # stay correct: eager from here on
                import warnings
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); the trainer continues with eager steps")
                self.usable = False
                self.cache.clear()
                torch.cuda.synchronize()
                self.eager += 1
                return r.run(pb, labels, device_step=True)
        else:
            _copy_into(ent["pb"], pb)
            if labels is not None:
                ent["labels"].copy_(labels, non_blocking=True)
            r.opt.global_step += 1  # (the replayed AdamW advances the DEVICE counter; the capture itself advanced the host's once)
        r.opt.sync_hyper()
        ent["graph"].replay()
        self.replays += 1
        return ent["out"]
