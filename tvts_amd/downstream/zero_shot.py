"""The device-side arithmetic of the zero-shot scripts (v2/downstream/zero_recognition_TVTSv2_ViT_B_16.py:67-108):
prompt-ensemble class embeddings and `100 * normalise(video) @ W` logits on HIP kernels, top-k accuracy."""
from __future__ import annotations

import torch

from .. import hip as K


def _l2norm(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous().float()
    xn, inv = torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    K.l2norm_rows(x, xn, inv, 0.0)  # the scripts divide by the plain norm (no eps clamp)
    return xn


@torch.no_grad()
def class_embedding(model, prompt_ids: torch.Tensor, n_patches: int) -> torch.Tensor:
    """One class: the tokenised prompts [P, ctx] go through the text tower (beside dummy frames, as the script does),
    are normalised, averaged and normalised again (:70-80)."""
    dev = model.store.device
    P = prompt_ids.shape[0]
    data = {"text": prompt_ids, "video": torch.zeros(P, 3, model.arch["image"], model.arch["image"], device=dev),
            "keep_ind": torch.arange(n_patches).unsqueeze(0)}
    emb, _ = model(data, return_embeds=True)
    mean = _l2norm(emb).mean(dim=0, keepdim=True)
    return _l2norm(mean)[0]


@torch.no_grad()
def class_logits(video_emb: torch.Tensor, zeroshot_weights: torch.Tensor) -> torch.Tensor:
    """100 * normalise(video_emb) @ zeroshot_weights (:99-100); zeroshot_weights is [E, n_classes]."""
    v = _l2norm(video_emb)
    w = zeroshot_weights.contiguous().float()
    E, C = w.shape
    out = torch.empty(v.shape[0], C, dtype=torch.float32, device=v.device)
    K.gemm_small(v, w, out, M=v.shape[0], N=C, K=E, sa=(E, 1), sb=(C, 1), alpha=100.0)
    return out


def accuracy(logits: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """Per k: the number of samples whose target class is among the k highest logits (what the scripts accumulate)."""
    own = logits.gather(1, target.view(-1, 1).long())
    higher = (logits > own).sum(dim=1)  # classes scoring above the target
    return [float((higher < k).sum()) for k in topk]
