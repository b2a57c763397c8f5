"""Drop-in for the retrieval metrics of v2/model/metric.py used by the pretraining validation step
(`config['metrics'] = ["t2v_metrics", "v2t_metrics"]`, train_dist_TVTSv2_ViT_B_16.py:64).

Same names, argument and result dict; the N x M comparisons that produce the ranks run in a HIP kernel
(`tvts_retrieval_ranks`), the seven summary numbers are taken from the rank vector exactly like `cols2metrics`."""
from __future__ import annotations

import numpy as np
import torch

from .. import hip as K


def _device_sims(sims) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise RuntimeError("tvts_amd metrics run on the GPU (no CPU fallback)")
    if not isinstance(sims, torch.Tensor):
        sims = torch.as_tensor(np.asarray(sims))
    return sims.detach().to("cuda", torch.float32).contiguous()


def cols2metrics(cols, num_queries):
    """v2/model/metric.py:285-296 on a rank vector (device tensor or array)."""
    cols = cols.detach().double().cpu().numpy() if isinstance(cols, torch.Tensor) else np.asarray(cols, dtype=np.float64)
    m = {"R1": 100 * float(np.sum(cols == 0)) / num_queries, "R5": 100 * float(np.sum(cols < 5)) / num_queries,
         "R10": 100 * float(np.sum(cols < 10)) / num_queries, "R50": 100 * float(np.sum(cols < 50)) / num_queries,
         "MedR": np.median(cols) + 1, "MeanR": np.mean(cols) + 1}
    stats = [m["R1"], m["R5"], m["R10"]]
    m["geometric_mean_R1-R5-R10"] = float(np.exp(np.mean(np.log(stats)))) if min(stats) > 0 else 0.0
    return m


def _valid_bytes(query_masks, n_text, device):
    if query_masks is None:
        return None
    qm = query_masks.detach().cpu().numpy() if isinstance(query_masks, torch.Tensor) else np.asarray(query_masks)
    assert qm.size == n_text, "invalid query mask shape"
    return torch.as_tensor(qm.reshape(-1) != 0).to(torch.uint8).to(device).contiguous()


def t2v_metrics(sims, query_masks=None):
    """sims[i, j] = <text_i, video_j>; ties broken optimistically; query_masks ([n_vid, captions_per_video], 0 = caption
    missing) drops those queries (metric.py:16-126)."""
    sims = _device_sims(sims)
    ranks = K.retrieval_ranks(sims, "t2v")
    valid = _valid_bytes(query_masks, sims.shape[0], sims.device)
    if valid is not None:
        ranks = ranks[valid.bool()]
    return cols2metrics(ranks, int(ranks.numel()))


def v2t_metrics(sims, query_masks=None):
    """closest own caption per video, ties averaged; missing captions are neither candidates nor targets (metric.py:129-187)."""
    sims = _device_sims(sims)
    ranks = K.retrieval_ranks(sims, "v2t", valid=_valid_bytes(query_masks, sims.shape[0], sims.device))
    return cols2metrics(ranks, sims.shape[1])
