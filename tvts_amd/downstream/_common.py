"""Inference model shared by downstream/model_TVTSv2_ViT_{B_16,B_32,H_14}.py (v2/downstream/model_TVTSv2_ViT_B_16.py:10-90)."""
from __future__ import annotations

import types

import torch

from ..arch import ARCHS
from ..model._common import TVTSv2Base, sim_matrix  # noqa: F401  (sim_matrix is imported from here by the zero_ret scripts)


class DownstreamBase(TVTSv2Base):
    """`TVTSv2_*(load_checkpoint=None)`: text tower + space-time ViT with mask ratio 0 and 12 temporal positions; the
    state dict has no `pred_model.*` keys (strict load of the released / fine-tuned checkpoints, `module.` prefix fixed
    like utils/util.py:25-50).  forward(data) -> (text_embeddings, video_embeddings), or their similarity matrix."""

    def __init__(self, load_checkpoint=None, arch=None, init_seed=0, pretrained=None):
        """pretrained: as in TVTSv2Base -- with an empty load_checkpoint the named classes initialise from the kept CLIP / OpenCLIP model
        like the reference's downstream constructors do (v2/downstream/model_TVTSv2_ViT_B_16.py:15-41) and raise when it is not there;
        False = random initialisation (tests), a mapping / path = that CLIP-layout state dict."""
        a = dict(arch if arch is not None else ARCHS[self.ARCH_NAME])
        a.update(mask_ratio=0.0, sort_head=False)
        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if load_checkpoint not in ["", None]:
            pretrained = False  # every tensor comes from the checkpoint's strict load below
        elif pretrained is None and arch is not None:
            pretrained = False
        super().__init__(types.SimpleNamespace(local_rank=dev, rank=0, world_size=1), load_checkpoint=None, arch=a,
                         init_seed=init_seed, pretrained="reference" if pretrained is None else pretrained)
        if load_checkpoint not in ["", None]:
            sd = torch.load(load_checkpoint, map_location=self.store.device, weights_only=False)["state_dict"]
            if next(iter(sd)).startswith("module."):
                sd = {k[7:]: v for k, v in sd.items()}
            self.load_state_dict(sd, strict=True)
            print("loading checkpoint from {}".format(load_checkpoint))
        for p in self.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def forward(self, data, return_embeds=True):
        te, ve, _ = super().forward(data, return_embeds=True)
        if return_embeds:
            return te, ve
        return sim_matrix(te, ve)
