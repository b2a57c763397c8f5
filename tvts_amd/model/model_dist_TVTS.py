"""Drop-in for v1/model/model_dist_TVTS.py: ``TVTS(args, video_params, text_params, projection_dim=256,
load_checkpoint=None, projection='minimal')`` -- DistilBERT text tower, tubelet ViT-B/16 with joint space-time attention,
ReLU+Linear / Linear projections, transcript-sorting head -- over the HIP step engine (tvts_amd/engine_v1.py).  Same
``forward(data, return_embeds=True) -> (text_embeds, video_embeds, pred_order)``, ``compute_text`` / ``compute_video``,
state-dict names and order, ``sim_matrix`` as an importable free function.

The text tower's dropout (Hugging Face DistilBERT, p = 0.1, active because of ``self.text_model.train()`` at
model_dist_TVTS.py:33-34) follows the module's training flag: ``model.train()`` steps draw counter-based masks
(tvts_amd/engine_v1.py), ``model.eval()`` (validation, compute_text for retrieval) runs without.  The pretrained initialisations (``AutoModel.from_pretrained``, ``./mae_pretrain_vit_base.pth``, :34,49-58) are
replaced by the same classes' random initialisers unless ``load_checkpoint`` names a checkpoint.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ..arch import ARCH_V1
from ..engine_v1 import EngineV1
from ._common import TVTSv2Base, sim_matrix  # noqa: F401


def reference_init_v1_(store, seed: int = 0):
    """DistilBERT: N(0, 0.02) weights / embeddings, zero biases, LayerNorm (1, 0) (transformers _init_weights); ViT: trunc-normal
    0.02 Linear weights, zero biases, LayerNorm (1, 0), cls / pos trunc-normal 0.02, temporal zeros, Conv3d default
    (v1/model/video_encoder.py:146-166); projections: nn.Linear default; sort head as in v2."""
    g = torch.Generator().manual_seed(seed)
    for name, shape in store.shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if "norm" in name.lower() and "type_embed" not in name:
            t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif name in ("video_model.temporal_embed", "pred_model.type_embed"):
            t = torch.zeros(shape)
        elif name.startswith("text_model.") or name in ("video_model.cls_token", "video_model.pos_embed"):
            t = torch.zeros(shape) if leaf == "bias" else torch.randn(shape, generator=g) * 0.02
        elif name.startswith("video_model.blocks."):
            t = torch.zeros(shape) if leaf == "bias" else (torch.randn(shape, generator=g) * 0.02).clamp_(-0.04, 0.04)
        elif leaf == "bias":
            w_shape = store.shapes[name[:-4] + "weight"]
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(int(np.prod(w_shape[1:])))
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        store.p(name).copy_(t.to(store.device))


class TVTS(TVTSv2Base):
    ENGINE = EngineV1
    INIT = staticmethod(reference_init_v1_)

    def __init__(self, args, video_params=None, text_params=None, projection_dim=256, load_checkpoint=None,
                 projection="minimal", arch=None, init_seed=0):
        video_params = video_params or {}
        text_params = text_params or {"model": "distilbert-base-uncased", "pretrained": True}
        if arch is None:
            if not text_params.get("pretrained", True):
                raise NotImplementedError("Huggingface text models require pretrained init.")  # model_dist_TVTS.py:31-32
            if not text_params.get("model", "distilbert-base-uncased").startswith("distilbert"):
                raise NotImplementedError("only the distilbert-base-uncased text tower of the shipped configs is built")
            if video_params.get("arch_config", "base_patch16_224") != "base_patch16_224" or projection != "minimal":
                raise NotImplementedError
            arch = dict(ARCH_V1, num_frames=video_params.get("num_frames", 16), embed=projection_dim)
        self.video_params, self.text_params = video_params, text_params
        super().__init__(args, load_checkpoint=load_checkpoint, arch=arch, init_seed=init_seed)

    def forward(self, data, return_embeds=True):
        self.engine.training = self.training  # dropout of the text tower in training mode only
        return super().forward(data, return_embeds)

    # pieces the reference exposes (model_dist_TVTS.py:131-147)
    def compute_text(self, text_data):
        self._fresh_shadows()
        eng = self.engine
        eng.training = self.training
        ids = text_data["input_ids"].detach().to("cpu", torch.int64)
        lens = text_data["attention_mask"].detach().to("cpu", torch.int64).sum(-1)
        N, L = ids.shape[0], int(lens.max())
        dev = self.store.device
        before, t = eng.text_forward_v1(ids[:, :L].to(torch.int32).contiguous().to(dev), lens.to(torch.int32).to(dev),
                                        (torch.arange(N) * L).to(torch.int32).to(dev), N, L)
        return before.clone(), t.clone()

    def compute_video(self, video_data, keep_ind):
        self._fresh_shadows()
        a = self.arch
        v = video_data.to(self.store.device, torch.float32).contiguous()
        B, T = v.shape[:2]
        tubes = T // a["tubelet"]
        keep = keep_ind[:, :tubes].to(torch.int32).contiguous().to(self.store.device)
        S = 1 + tubes * keep.shape[2]
        out, emb = self.engine.video_forward_v1(v, keep, B, tubes, (torch.arange(B) * S).to(torch.int32).to(self.store.device))
        return out.view(B, S, -1).clone(), emb.clone()
