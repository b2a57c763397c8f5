"""Data-parallel exchange steps of the TVTSv2 step: one process per GPU, torch.distributed (RCCL on ROCm).

* ``allgather_embeds`` -- the reference's ``AllGather_multi`` (v2/trainer/trainer.py:41-57): forward gathers
  the per-rank ``[B,E]`` embeddings to ``[W*B,E]``; backward is a LOCAL ROW SLICE of the incoming gradient,
  with no collective (``local_rows``).  Video and text embeddings travel in one fused ``[B,2E]`` message.
* ``GradSync`` -- the DDP gradient average (v2/base/base_trainer.py:23-25) over the flat fp32 gradient
  buffer: ranges are all-reduced (SUM) asynchronously as soon as the hand-written backward has finished
  them, overlapping RCCL traffic over xGMI with the remaining backward GEMMs; the 1/W factor is folded
  into the AdamW kernel.
These functions only move tensors; they run on gloo/CPU tensors in the tests exactly as on RCCL.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def allgather_embeds(video_emb: torch.Tensor, text_emb: torch.Tensor, scratch=None):
    """-> (video_all [W*B,E], text_all [W*B,E]); rank r owns rows r*B..(r+1)*B-1."""
    W, _ = world()
    if W == 1:
        return video_emb, text_emb
    B, E = video_emb.shape
    packed = torch.cat([video_emb, text_emb], dim=1).contiguous()
    out = torch.empty(W * B, 2 * E, dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    return out[:, :E].contiguous(), out[:, E:].contiguous()


class EmbedGather:
    """The same exchange started early: `start` is called by the engine as soon as both embeddings exist (before the
    sort head's forward), the collective runs asynchronously on RCCL's stream, `result` waits for it where the loss needs
    the gathered rows (trainer.py:479-483)."""

    def __init__(self):
        self.W, _ = world()
        self.pending = None

    def start(self, text_emb: torch.Tensor, video_emb: torch.Tensor):
        if self.W == 1:
            self.pending = (None, video_emb, text_emb)
            return
        B, E = video_emb.shape
        packed = torch.cat([video_emb, text_emb], dim=1).contiguous()
        out = torch.empty(self.W * B, 2 * E, dtype=packed.dtype, device=packed.device)
        self.pending = (dist.all_gather_into_tensor(out, packed, async_op=True), out, E)

    def result(self):
        h, a, b = self.pending
        self.pending = None
        if h is None:
            return a, b
        h.wait()
        return a[:, :b].contiguous(), a[:, b:].contiguous()


def local_rows(grad_all: torch.Tensor, B: int) -> torch.Tensor:
    """AllGather_multi.backward: this rank's rows of the gradient wrt the gathered tensor."""
    _, r = world()
    return grad_all[B * r:B * (r + 1)]


class GradSync:
    """Asynchronous bucketed all-reduce of ranges of one flat gradient buffer."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20):
        self.g = flat_grad
        self.W, _ = world()
        self.handles: List = []
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())

    def reduce_range(self, start: int, end: int):
        """Called once the backward no longer writes grad[start:end]."""
        if self.W == 1 or end <= start:
            return
        for s in range(start, end, self.bucket_elems):
            e = min(end, s + self.bucket_elems)
            self.handles.append(dist.all_reduce(self.g[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> float:
        """Wait for outstanding reductions; returns the scale (1/W) still to be applied to the sum."""
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / self.W
