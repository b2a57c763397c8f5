"""Synthetic loader honouring the reference batch-dict contract:

  video    fp32 [B,T,3,H,W]        (YTTemporal_dataset.py:200-245 / base_dataset.py:102-142, post-normalisation scale)
  text     int32 [NT*B, context]    clip-major rows i*B+b, SOT at 0, EOT = row max, zero pad (CLIP/clip/clip.py:220-235)
  keep_ind int64 [B,n]              unsorted prefix of a permutation, shared by all frames (YTTemporal_dataset.py:207-213)
  label    int64 [B,4] = arange(4)  (:149); absent for WebVid-style NT=1 batches
Exposes the attributes the trainer reads from a data loader (batch_size, n_samples, dataset_name, train_sampler).
"""
from __future__ import annotations

import torch

from ..arch import n_keep, patches_per_frame


def synth_batch(arch, B, T, seed=0, n_trans=None, caption_len=32, device="cpu"):
    NT = arch["n_trans"] if n_trans is None else n_trans
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, T, 3, arch["image"], arch["image"], generator=g, dtype=torch.float32)
    L = arch["context"]
    cl = min(caption_len, L)
    text = torch.zeros(NT * B, L, dtype=torch.int32)
    sot, eot = arch["vocab"] - 2, arch["vocab"] - 1
    text[:, 0] = sot
    hi = arch["vocab"] - 408 if arch["vocab"] > 1000 else arch["vocab"] - 2
    text[:, 1:cl - 1] = torch.randint(1, hi, (NT * B, cl - 2), generator=g, dtype=torch.int32)
    text[:, cl - 1] = eot
    ppf, n = patches_per_frame(arch), n_keep(arch)
    keep = torch.stack([torch.randperm(ppf, generator=g)[:n] for _ in range(B)]).to(torch.int64)
    batch = {"video": video.to(device), "text": text, "keep_ind": keep}
    if NT == arch["n_trans"]:
        batch["label"] = torch.arange(arch["n_trans"]).repeat(B, 1)
    return batch


def synth_batch_v1(arch, B, T, seed=0, n_trans=None, caption_len=32, device="cpu"):
    """v1 (TVTS) batch dict: text = the Hugging Face tokenizer's {'input_ids', 'attention_mask'} [NT*B, L] right-padded to the
    longest caption of the batch, [CLS] 101 first / [SEP] 102 last (v1/trainer/trainer.py:121-131); keep_ind [B, n_tubes, n_keep]
    with one tube mask PER TUBE (v1/data_loader/YTTemporal_dataset.py:207-215)."""
    NT = arch["n_trans"] if n_trans is None else n_trans
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, T, 3, arch["image"], arch["image"], generator=g, dtype=torch.float32)
    N, L = NT * B, caption_len
    lens = torch.randint(max(3, L // 2), L + 1, (N,), generator=g)
    lens[0] = L
    pos = torch.arange(L)[None, :]
    mask = (pos < lens[:, None]).to(torch.int64)
    cls_id, sep_id = (101, 102) if arch["vocab"] > 1000 else (arch["vocab"] - 2, arch["vocab"] - 1)
    ids = torch.randint(1, min(arch["vocab"], 30000) - 2, (N, L), generator=g) * mask
    ids[:, 0] = cls_id
    ids[torch.arange(N), lens - 1] = sep_id
    ppf, nk, tubes = patches_per_frame(arch), n_keep(arch), T // arch["tubelet"]
    keep = torch.stack([torch.stack([torch.randperm(ppf, generator=g)[:nk] for _ in range(tubes)]) for _ in range(B)])
    batch = {"video": video.to(device), "text": {"input_ids": ids, "attention_mask": mask}, "keep_ind": keep.to(torch.int64)}
    if NT == arch["n_trans"]:
        batch["label"] = torch.arange(arch["n_trans"]).repeat(B, 1)
    return batch


class _Sampler:
    def set_epoch(self, epoch):
        self.epoch = epoch


class SyntheticTextVideoLoader:
    def __init__(self, arch, batch_size, num_frames, n_batches, dataset_name="YTTemporal", n_trans=None, seed=0,
                 caption_len=32, device="cpu"):
        self.arch, self.batch_size, self.T, self.n_batches = arch, batch_size, num_frames, n_batches
        self.dataset_name, self.n_trans, self.seed, self.caption_len, self.device = dataset_name, n_trans, seed, caption_len, device
        self.n_samples = batch_size * n_batches
        self.train_sampler = _Sampler()

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for i in range(self.n_batches):
            yield synth_batch(self.arch, self.batch_size, self.T, seed=self.seed + i, n_trans=self.n_trans,
                              caption_len=self.caption_len, device=self.device)
