"""Drop-in for v2/trainer/trainer.py: Trainer_TVTSv2_{B_32,B_16,H_14} with the reference's constructor and
``train()`` contract, running each optimisation step on the HIP engine (tvts_amd.step.StepRunner).

Kept from the reference's ``_train_epoch`` (:419-525): the YT loader drives the loop, the other loaders are
cycled with fresh iterators (:438-461); per loop iteration one optimizer step per loader (:463); captions
arrive as a list of NT lists and are flattened clip-major before tokenising (:465-473, ``truncate=True`` for
the B models, none for H/14 :767); loss = InfoNCE + 2*CE (:485-496); the debug log line and its cadence
``int(sqrt(batch_size))`` (:384,505-512); LR x0.1 at the ``--schedule`` epochs, applied at epoch end (:402-417).
AllGather / AllGather_multi are exported with the same names and semantics for callers that import them.
"""
from __future__ import annotations

import collections.abc

import numpy as np
import torch
import torch.distributed as dist

from ..base import Multi_BaseTrainer_dist
from ..model._common import sim_matrix  # noqa: F401  (the reference re-exports it at module scope)
from ..step import GraphReplay, StepRunner

__all__ = ["AllGather", "AllGather_multi", "Trainer_TVTSv2_B_32", "Trainer_TVTSv2_B_16", "Trainer_TVTSv2_H_14", "Trainer_TVTS"]


class AllGather_multi(torch.autograd.Function):
    """all_gather forward, local-row-slice backward, no reduction (v2/trainer/trainer.py:41-57)."""

    @staticmethod
    def forward(ctx, tensor, n_gpu, args):
        ctx.rank, ctx.batch_size = args.rank, tensor.shape[0]
        if args.world_size == 1:
            return tensor.clone()
        out = torch.empty((args.world_size * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                          device=tensor.device)
        dist.all_gather_into_tensor(out, tensor.contiguous())
        return out

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None, None


class AllGather(AllGather_multi):
    """Single-node variant (:22-38): identical except that it indexes with local_rank."""

    @staticmethod
    def forward(ctx, tensor, n_gpu, args):
        out = AllGather_multi.forward(ctx, tensor, n_gpu, args)
        ctx.rank = args.local_rank
        return out


class _Cycled:
    """A loader that starts over (with a fresh iterator, i.e. a fresh shuffle) whenever it runs dry."""

    def __init__(self, loader):
        self.loader, self.it = loader, iter(loader)

    def __next__(self):
        for _ in range(2):
            try:
                return next(self.it)
            except StopIteration:
                self.it = iter(self.loader)
        raise RuntimeError("empty data loader")


def _lockstep(loaders, len_epoch):
    """One list of batches (one per loader, in loader order) per loop iteration.  The first loader whose length is the epoch
    length paces the epoch and is iterated exactly once; every other loader is cycled beside it (v2/trainer/trainer.py:438-461:
    the YT-Temporal loader drives, the WebVid / CC3M loaders restart as often as needed)."""
    pace = next((i for i, dl in enumerate(loaders) if len(dl) == len_epoch), None)
    if pace is None:
        raise ValueError(f"no data loader has the epoch length {len_epoch}")
    others = {i: _Cycled(dl) for i, dl in enumerate(loaders) if i != pace}
    for lead_batch in loaders[pace]:
        yield [lead_batch if i == pace else next(others[i]) for i in range(len(loaders))]


class _LazyLog:
    """The debug line of the logging steps (v2/trainer/trainer.py:505-512) without stopping the host: the step's two losses are copied
    to page-locked memory behind the step (asynchronously, one event per line) and the line is written as soon as its event has
    completed -- checked at every later step, forced at the end of the epoch.  Same lines, same order, same digits (the total is
    the fp32 sum of the two fp32 losses, as `loss1 + loss2` is on the device); a line appears a step or two later than the
    reference's `.item()` would print it.  With the reference's cadence int(sqrt(batch_size)) -- every 3rd step at 12 pairs per GPU --
    three blocking reads per line drained the launch queue every third step: 16.0 -> 15.x ms per step at 12 pairs."""

    SLOTS = 64

    def __init__(self, logger, fmt, device):
        self.logger, self.fmt, self.q, self.i = logger, fmt, [], 0
        self.cuda = torch.cuda.is_available() and torch.device(device).type == "cuda"
        self.host = torch.zeros(self.SLOTS, 2, dtype=torch.float32, pin_memory=self.cuda)

    def add(self, head, l1, l2):
        if len(self.q) >= self.SLOTS:
            self.flush(block=True)
        slot = self.host[self.i]
        self.i = (self.i + 1) % self.SLOTS
        slot[0:1].copy_(l1.reshape(1), non_blocking=True)
        if l2 is None:
            slot[1] = 0.0
        else:
            slot[1:2].copy_(l2.reshape(1), non_blocking=True)
        ev = None
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
        self.q.append((ev, slot, head))

    def flush(self, block=False):
        while self.q and (block or self.q[0][0] is None or self.q[0][0].query()):
            ev, slot, head = self.q.pop(0)
            if ev is not None:
                ev.synchronize()
            a, b = np.float32(slot[0].item()), np.float32(slot[1].item())
            self.logger.debug(self.fmt.format(*head, float(a), float(b), float(np.float32(a + b))))


class _TrainerBase(Multi_BaseTrainer_dist):
    TRUNCATE = True
    CACHE_CAPTIONS = True
    LOG_LINE = "Train Epoch: {} dl{} {} Loss_ct: {:.6f} Loss_ce: {:.6f} Loss: {:.6f}"  # v2/trainer/trainer.py:505-512

    def __init__(self, args, model, loss, metrics, optimizer, config, data_loader, valid_data_loader=None,
                 lr_scheduler=None, len_epoch=None, writer=None, visualizer=None, tokenizer=None,
                 max_samples_per_epoch=50000):
        super().__init__(args, model, loss, metrics, optimizer, config, writer)
        self.config, self.args = config, args
        self.data_loader = data_loader
        if len_epoch is None:
            self.len_epoch = None
            for x in data_loader:
                if x.dataset_name.startswith("YT"):
                    self.len_epoch = len(x)
            if self.len_epoch is None:
                self.len_epoch = len(data_loader[0])
        else:
            self.len_epoch = len_epoch
        self.valid_data_loader = valid_data_loader
        self.do_validation = bool(valid_data_loader)
        self.lr_scheduler, self.visualizer = lr_scheduler, visualizer
        self.batch_size = self.data_loader[0].batch_size
        self.log_step = max(1, int(np.sqrt(self.batch_size)))
        self.total_batch_sum = sum(x.batch_size for x in self.data_loader)
        # tokenised-caption cache keyed by caption text: rows identical to calling the tokenizer directly
        from ..data_loader.transforms import CaptionCache
        self.tokenizer = CaptionCache(tokenizer) if (tokenizer is not None and self.CACHE_CAPTIONS) else tokenizer
        self.max_samples_per_epoch = max_samples_per_epoch
        self.n_gpu = self.args.world_size
        self.allgather = AllGather_multi.apply
        self.num_clips = self.n_trans = 4
        self.base_lr = [g["lr"] for g in optimizer.param_groups]
        # the configured loss module's temperature drives the fused loss head (model/loss.py:11 NormSoftmaxLoss(temperature))
        from ..engine import LossHead
        self.runner = StepRunner(model, optimizer, LossHead(model.store.device, temperature=float(getattr(loss, "temperature", 0.05))))
        # TVTS_TRAINER_GRAPH=1: batches of a repeating shape are replayed from a captured hipGraph (world 1, fused optimizer).  OPT-IN:
        # measured equal to slower than the eager loop once prepare_batch stopped synchronising (profiles/r06_bench_product_path*.txt:
        # 141.8 - 142.2 against 140.3 - 140.5 ms at 192 pairs, 16.4 against 15.7 - 16.0 ms at 12) -- the copy into the captured
        # buffers and the replay of a two-stream graph cost what the saved launches return
        self.replay = GraphReplay(self.runner)

    def _adjust_learning_rate(self, optimizer, epoch, args):
        lr_rate = 1.0
        for milestone in getattr(args, "schedule", []):
            if epoch == milestone:
                lr_rate = 0.1
        for group in optimizer.param_groups:
            group["lr"] = group["lr"] * lr_rate

    def _progress(self, batch_idx, dl_idx):
        current = batch_idx * self.data_loader[dl_idx].batch_size
        total = getattr(self.data_loader[dl_idx], "n_samples", self.len_epoch * self.data_loader[dl_idx].batch_size)
        return "[{}/{} ({:.0f}%)]".format(current, total, 100.0 * current / max(total, 1))

    def _tokenize(self, data):
        if self.tokenizer is not None and not torch.is_tensor(data["text"]):
            text_all = []
            for clip_texts in data["text"]:  # list over the NT clips, each a list of B strings -> clip-major
                text_all = text_all + list(clip_texts)
            data["text"] = self.tokenizer(text_all, truncate=True) if self.TRUNCATE else self.tokenizer(text_all)
        return data

    def _train_epoch(self, epoch):
        total_loss = [0.0] * len(self.data_loader)
        # every step's loss is kept on the device and read at the end of the epoch (and on the logging steps): the reference's
        # per-step `.item()` (trainer.py:503) would stop the host at every step, and with it the next batch's host-to-device copy
        # that otherwise travels under the running step (Engine._clip_to_device) -- the logged and returned values are the same
        step_loss = torch.zeros(len(self.data_loader), self.len_epoch, dtype=torch.float32, device=self.model.store.device)
        for loader in self.data_loader:
            if hasattr(loader, "train_sampler") and loader.train_sampler is not None:
                loader.train_sampler.set_epoch(epoch)
        lazy = _LazyLog(self.logger, self.LOG_LINE, self.model.store.device)
        for batch_idx, data_li in enumerate(_lockstep(self.data_loader, self.len_epoch)):
            for dl_idx, data in enumerate(data_li):
                out = self.replay.step(self._tokenize(data))
                log_now = batch_idx % self.log_step == 0 and self.args.local_rank == 0
                l1 = out["loss1"]
                l2 = out["loss2"]
                if batch_idx < self.len_epoch:
                    if l2 is None:
                        step_loss[dl_idx, batch_idx].copy_(l1.reshape(()))
                    else:
                        torch.add(l1.reshape(()), l2.reshape(()), out=step_loss[dl_idx, batch_idx])
                else:
                    total_loss[dl_idx] += float(l1 if l2 is None else l1 + l2)
                if log_now:
                    lazy.add((epoch, dl_idx, self._progress(batch_idx, dl_idx)), l1, l2)
                lazy.flush()
        lazy.flush(block=True)
            # max_samples_per_epoch is stored and never read by the reference's TVTSv2 trainers (trainer.py:93,387,681):
            # the whole YT loader is iterated, so the per-epoch LR schedule sees the same number of steps here
        for dl_idx, row in enumerate(step_loss.cpu().tolist()):
            total_loss[dl_idx] += sum(row)  # the same per-step fp32 values the reference adds up as Python floats
        log = {f"loss_{dl_idx}": total_loss[dl_idx] / self.len_epoch for dl_idx in range(len(self.data_loader))}
        if self.do_validation:
            val_log = self._valid_epoch(epoch)
            if self.args.rank == 0:
                log.update(val_log)
        self._adjust_learning_rate(self.optimizer, epoch, self.args)
        return log

    @torch.no_grad()
    def _valid_epoch(self, epoch):
        """Validation after an epoch (v2/trainer/trainer.py:527-635): eval forward on the HIP engine, embeddings /
        labels / predicted orders gathered over the ranks, sorting accuracy (a sample counts when all of its NT
        positions are right, :575-583), and the configured retrieval metrics on the text x video similarity matrix
        of everything seen -- similarity and ranks computed on the device (tvts_amd.model.metric)."""
        from ..model._common import sim_matrix
        self.model.eval()
        world = self.args.world_size if dist.is_available() and dist.is_initialized() else 1

        def gather(t):
            if world == 1:
                return t
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            return torch.cat(parts, dim=0)

        res_dict, nested = {}, {}
        for dl_idx, dl in enumerate(self.valid_data_loader):
            hits, count = 0, 0
            text_arr, vid_arr = [], []
            for data in dl:
                data = self._tokenize(data)
                te, ve, pred = self.model(data, return_embeds=True)
                text_arr.append(gather(te.detach().float()))
                vid_arr.append(gather(ve.detach().float()))
                if pred is not None:
                    labels_all = gather(data["label"].to(pred.device))
                    preds_all = gather(pred.argmax(dim=-1))
                    hits += int((preds_all == labels_all).all(dim=1).sum())
                    count += preds_all.shape[0]
                else:  # the reference reports 1 / 1 for loaders without transcripts (:590-592)
                    hits, count = 1, 1
            sims = sim_matrix(torch.cat(text_arr), torch.cat(vid_arr))
            name = getattr(dl, "dataset_name", f"val{dl_idx}")
            nested[dl_idx] = {}
            for metric in self.metrics:
                res = metric(sims)
                nested[dl_idx][metric.__name__] = res
                if self.args.rank == 0:
                    verbose(epoch=epoch, metrics=res, name=name, mode=metric.__name__)
                    if self.writer is not None:
                        for key, val in format_nested_metrics_for_writer(res, mode=metric.__name__, name=name).items():
                            self.writer.log_scalar(key, val)
            if self.writer is not None and self.args.rank == 0:
                self.writer.log_scalar(f"loss_val_{dl_idx}", hits / max(len(dl), 1))
            if self.args.rank == 0:
                res_dict[f"val_loss_{dl_idx}"] = hits / count  # (sic) the reference logs the accuracy under this key (:630)
        if self.args.rank == 0 and "val_loss_0" in res_dict:
            print("Top-1 Accuracy for Frame Prediction:", res_dict["val_loss_0"])
        self.last_val_metrics = nested
        self.model.train()
        return res_dict


def verbose(epoch, metrics, mode, name="TEST"):
    """The one-line retrieval summary the reference prints per loader and metric (v2/trainer/trainer.py:942-947): same text."""
    recalls = ", ".join(f"R@{k}: {metrics['R%d' % k]:.1f}" for k in (1, 5, 10, 50))
    print(f"[{mode}]{name:s} epoch {epoch}, {recalls}, MedR: {metrics['MedR']:g}, MeanR: {metrics['MeanR']:.1f}")


def format_nested_metrics_for_writer(metrics, mode, name="TEST"):
    """Scalar names of the writer: ``[<metric fn>]<loader>_<statistic>`` (v2/trainer/trainer.py:950-955)."""
    tag = f"[{mode}]{name}_"
    return {tag + stat: value for stat, value in metrics.items()}


class Trainer_TVTSv2_B_32(_TrainerBase):
    TRUNCATE = True


class Trainer_TVTSv2_B_16(_TrainerBase):
    TRUNCATE = True


class Trainer_TVTSv2_H_14(_TrainerBase):
    TRUNCATE = False


class Trainer_TVTS(_TrainerBase):
    """v1/trainer/trainer.py:40-310: the same epoch loop over the v1 model.  Differences kept from the
    reference: the epoch follows the LONGEST loader (:50-54), captions go through the Hugging Face tokenizer call
    ``tokenizer(text, return_tensors='pt', padding=True, truncation=True, max_length=50)`` (:121-131), the debug line says
    Loss_align / Loss_sort (:164-172), and the learning rate is recomputed from the base rate at every epoch end --
    lr = base_lr * 0.1 ** (number of milestones reached) (:80-91) -- instead of being decayed in place."""
    LOG_LINE = "Train Epoch: {} dl{} {} Loss_align: {:.6f} Loss_sort: {:.6f} Loss: {:.6f}"
    CACHE_CAPTIONS = False  # the Hugging Face tokenizer pads to the longest caption of the BATCH: rows are not per-caption constants

    def __init__(self, args, model, loss, metrics, optimizer, config, data_loader, valid_data_loader=None, lr_scheduler=None,
                 len_epoch=None, writer=None, visualizer=None, tokenizer=None, max_samples_per_epoch=50000):
        if len_epoch is None:
            len_epoch = max(len(x) for x in data_loader)
        super().__init__(args, model, loss, metrics, optimizer, config, data_loader, valid_data_loader, lr_scheduler, len_epoch,
                         writer, visualizer, tokenizer, max_samples_per_epoch)
        self.base_lr = optimizer.state_dict()["param_groups"][0]["lr"]  # v1/base/base_trainer.py:30
        # the Hugging Face tokenizer pads to the longest caption of the BATCH: the caption length -- and with it the shape of the
        # step -- changes from batch to batch, so there is no repeating signature to capture
        self.replay.usable = False

    # the dropout generator's state travels with the checkpoint: a resumed run draws the masks the uninterrupted run would have,
    # on EVERY rank -- the file (written by rank 0) holds the rank-independent part of the seed, each rank adds its own offset back
    def _extra_state(self):
        eng = self.model.engine
        return {"drop_seed_base": eng.drop_seed_base()} if hasattr(eng, "drop_seed") else None

    def _load_extra_state(self, state):
        eng = self.model.engine
        if not state or not hasattr(eng, "drop_seed"):
            return
        if "drop_seed_base" in state:
            eng.set_drop_seed_base(int(state["drop_seed_base"]))
        elif "drop_seed" in state:  # a round-4 checkpoint: rank 0's full seed (its rank offset is 0, so this IS the base)
            eng.set_drop_seed_base(int(state["drop_seed"]))

    def _adjust_learning_rate(self, optimizer, epoch, args):
        lr = self.base_lr
        for milestone in getattr(args, "schedule", []):
            lr *= 0.1 if epoch >= milestone else 1.0
        for group in optimizer.param_groups:
            group["lr"] = lr

    def _tokenize(self, data):
        # an already tokenised batch (a dict or a Hugging Face BatchEncoding -- a UserDict, not a dict) passes through
        if self.tokenizer is not None and not (isinstance(data["text"], collections.abc.Mapping) or hasattr(data["text"], "input_ids")):
            text_all = []
            for clip_texts in data["text"]:
                text_all = text_all + list(clip_texts)
            data["text"] = self.tokenizer(text_all, return_tensors="pt", padding=True, truncation=True, max_length=50)
        return data
