#!/usr/bin/env python3
"""Generate the v1 (TVTS) parity fixtures by RUNNING THE REFERENCE (build container only).

/root/reference/v1 is imported read-only with import shims for what this image lacks (timm) or cannot fetch (pretrained
DistilBERT / MAE weights: `AutoModel.from_pretrained` and the `./mae_pretrain_vit_base.pth` load are replaced by seeded
random initialisation of the SAME classes -- transformers' DistilBertModel, the reference's own VisionTransformer).  No
reference file is edited or copied; only data (inputs are regenerated from seeds, expected outputs / gradients are
stored) lands in tests/golden/v1_*.npz.

    python tests/golden/make_golden_v1.py [tiny] [dropout] [full]
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TVTS_REFERENCE_V1", "/root/reference/v1")
sys.path.insert(0, ROOT)

from oracle import tvts_v1_oracle as V  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def import_reference_v1():
    if "ns" in globals():
        return globals()["ns"]
    import transformers
    from transformers import DistilBertConfig, DistilBertModel
    _stub("timm"); _stub("timm.models")
    _stub("timm.models.layers", StdConv2dSame=object, DropPath=torch.nn.Identity,
          to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x), trunc_normal_=lambda t, std=1.0: t)
    _stub("humanize")
    base = _stub("base")
    base.BaseModel = _load("base.base_model", os.path.join(REF, "base/base_model.py")).BaseModel
    utils = _stub("utils")
    utils.util = _load("utils.util", os.path.join(REF, "utils/util.py"))
    model_pkg = _stub("model")
    st = _load("model.sort_transformer", os.path.join(REF, "model/sort_transformer.py"))
    ve = _load("model.video_encoder", os.path.join(REF, "model/video_encoder.py"))
    model_pkg.sort_transformer, model_pkg.video_encoder = st, ve
    loss = _load("model.loss", os.path.join(REF, "model/loss.py"))

    # pretrained weights are not in the image: the SAME classes with seeded random weights, dropout 0
    def fake_from_pretrained(name, *a, **k):
        assert name == "distilbert-base-uncased"
        torch.manual_seed(4321)
        return DistilBertModel(DistilBertConfig(dropout=0.0, attention_dropout=0.0))
    transformers.AutoModel.from_pretrained = staticmethod(fake_from_pretrained)
    real_load = torch.load

    def fake_load(path, *a, **k):
        if str(path).endswith("mae_pretrain_vit_base.pth"):
            return {"model": {}}
        return real_load(path, *a, **k)
    torch.load = fake_load
    tv = _load("model.model_dist_TVTS", os.path.join(REF, "model/model_dist_TVTS.py"))
    ns = types.SimpleNamespace(tvts=tv, ve=ve, sort=st, loss=loss, DistilBertConfig=DistilBertConfig, DistilBertModel=DistilBertModel)
    globals()["ns"] = ns
    return ns


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_losses(ns, te, ve, pred, label):
    # v1/trainer/trainer.py:142-152
    loss1 = ns.loss.NormSoftmaxLoss()(ns.tvts.sim_matrix(ve, te))
    loss2 = torch.nn.CrossEntropyLoss()(pred.reshape(-1, pred.shape[-1]), label.reshape(-1)) * 2
    return loss1, loss2


class TinyV1(torch.nn.Module):
    """The reference's sub-modules at the tiny architecture, wired by the reference's own TVTS.forward / compute_text /
    compute_video (called as unbound functions)."""

    def __init__(self, ns, a, P, dropout=0.0):
        super().__init__()
        from functools import partial
        self.text_params = {"model": "distilbert-base-uncased"}
        cfg = ns.DistilBertConfig(
            vocab_size=a["vocab"], max_position_embeddings=a["max_pos"], n_layers=a["text_layers"], n_heads=a["text_heads"],
            dim=a["text_width"], hidden_dim=a["text_ffn"], dropout=dropout, attention_dropout=dropout)
        if dropout > 0:
            cfg._attn_implementation = "eager"  # the attention path that calls nn.functional.dropout on the probabilities
        self.text_model = ns.DistilBertModel(cfg)
        vm = ns.ve.VisionTransformer(img_size=a["image"], patch_size=a["patch"], embed_dim=a["width"], depth=a["layers"],
                                     num_heads=a["heads"], mlp_ratio=4, qkv_bias=True,
                                     norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=a["num_frames"],
                                     tubelet_size=a["tubelet"])
        vm.pre_logits = torch.nn.Identity()
        self.video_model = vm
        self.txt_proj = torch.nn.Sequential(torch.nn.ReLU(), torch.nn.Linear(a["text_width"], a["embed"]))
        self.vid_proj = torch.nn.Sequential(torch.nn.Linear(a["width"], a["embed"]))
        self.n_trans = a["n_trans"]
        self.pred_model = ns.sort.SortTransformer(num_classes=a["n_trans"], embed_dim=a["sort_width"], num_heads=a["sort_heads"],
                                                  depth=a["sort_depth"])
        self.load_state_dict(P, strict=True)
        self._ref = ns.tvts.TVTS

    def compute_text(self, t):
        return self._ref.compute_text(self, t)

    def compute_video(self, v, k):
        return self._ref.compute_video(self, v, k)

    def forward(self, data, return_embeds=True):
        return self._ref.forward(self, data, return_embeds)


def _run(model, batch, ns, sel, name, extra):
    model.train()
    te, ve, pred = model(batch)
    loss1, loss2 = ref_losses(ns, te, ve, pred, batch["label"])
    (loss1 + loss2).backward()
    pd = dict(model.named_parameters())
    names = [k for k, p in pd.items() if p.grad is not None]
    gn_vals = np.array([float(pd[k].grad.norm()) for k in names], dtype=np.float64)
    grads = {}
    for key, (pname, idx) in sel.items():
        grads[key] = pd[pname].grad[idx]
    save(name, te=te, ve=ve, pred=pred, loss1=loss1, loss2=loss2, grad_norm=float(np.sqrt((gn_vals ** 2).sum())),
         gn_names=np.array(names), gn_vals=gn_vals, param_names=np.array(list(model.state_dict().keys())), **grads, **extra)


def gen_tiny():
    ns = import_reference_v1()
    a = V.tiny_arch()
    P = V.synth_params(a, seed=31)
    m = TinyV1(ns, a, P)
    assert list(m.state_dict().keys()) == list(V.param_shapes(a).keys())
    batch = V.synth_batch(a, B=3, T=4, seed=32, caption_len=11)
    sel = {"g_conv": ("video_model.patch_embed.proj.weight", (slice(0, 4),)),
           "g_pos": ("video_model.pos_embed", (slice(None),)),
           "g_temporal": ("video_model.temporal_embed", (slice(None),)),
           "g_cls": ("video_model.cls_token", (slice(None),)),
           "g_qkv1": ("video_model.blocks.1.attn.qkv.weight", (slice(0, 16),)),
           "g_word": ("text_model.embeddings.word_embeddings.weight", (slice(None),)),
           "g_posemb": ("text_model.embeddings.position_embeddings.weight", (slice(None),)),
           "g_qlin0": ("text_model.transformer.layer.0.attention.q_lin.weight", (slice(None),)),
           "g_lin2": ("text_model.transformer.layer.1.ffn.lin2.weight", (slice(None),)),
           "g_txtproj": ("txt_proj.1.weight", (slice(None),)),
           "g_vidproj": ("vid_proj.0.weight", (slice(None),)),
           "g_head": ("pred_model.head.weight", (slice(None),))}
    _run(m, batch, ns, sel, "v1_tiny", dict(seed=31, batch_seed=32, B=3, T=4, caption_len=11))
    # WebVid-style batch: one caption per video, no sorting head (model_dist_TVTS.py:113-116)
    b1 = V.synth_batch(a, B=3, T=4, seed=33, n_trans=1, caption_len=9)
    te, ve, pred = m(b1)
    assert pred is None
    save("v1_tiny_nt1", te=te, ve=ve, loss1=ns.loss.NormSoftmaxLoss()(ns.tvts.sim_matrix(ve, te)), seed=31, batch_seed=33)


def gen_tiny_dropout():
    """The reference's training-mode text tower: the REAL transformers DistilBertModel with dropout = attention_dropout = 0.1 in
    train() mode (v1/model/model_dist_TVTS.py:33-34), wired by the reference's own TVTS.forward.  torch.nn.functional.dropout --
    what nn.Dropout and the eager attention path call -- is replaced for the duration of the forward by the build's counter-based
    mask generator (oracle drop_mask; call k of the forward = site k), so that the fixture pins WHERE the reference drops and how
    it scales, independent of torch's random stream.  The captions are cut to the longest one first (the build's batch format)."""
    ns = import_reference_v1()
    a = V.tiny_arch()
    P = V.synth_params(a, seed=31)
    p_drop, seed = 0.1, 0x5EED5EED
    m = TinyV1(ns, a, P, dropout=p_drop)
    batch = V.synth_batch(a, B=3, T=4, seed=34, caption_len=11)
    batch = dict(batch, text=V.trim_text(batch["text"]))
    import torch.nn.functional as F
    real, calls = F.dropout, []

    def counter_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        assert abs(p - p_drop) < 1e-12
        site = len(calls)
        calls.append(tuple(x.shape))
        return x * V.drop_mask(seed, site, tuple(x.shape), p)
    F.dropout = counter_dropout
    torch.nn.functional.dropout = counter_dropout
    try:
        sel = {"g_word": ("text_model.embeddings.word_embeddings.weight", (slice(None),)),
               "g_qlin0": ("text_model.transformer.layer.0.attention.q_lin.weight", (slice(None),)),
               "g_vlin1": ("text_model.transformer.layer.1.attention.v_lin.weight", (slice(None),)),
               "g_lin2": ("text_model.transformer.layer.1.ffn.lin2.weight", (slice(None),)),
               "g_lin1": ("text_model.transformer.layer.0.ffn.lin1.weight", (slice(None),)),
               "g_txtproj": ("txt_proj.1.weight", (slice(None),)),
               "g_head": ("pred_model.head.weight", (slice(None),))}
        _run(m, batch, ns, sel, "v1_tiny_dropout", dict(seed=31, batch_seed=34, B=3, T=4, caption_len=11, p=p_drop, drop_seed=seed,
                                                       n_dropout_calls=1 + 2 * a["text_layers"]))
    finally:
        F.dropout = real
        torch.nn.functional.dropout = real
    N, L, h, W = batch["text"]["input_ids"].shape[0], batch["text"]["input_ids"].shape[1], a["text_heads"], a["text_width"]
    want = [(N, L, W)] + [(N, h, L, L), (N, L, W)] * a["text_layers"]
    assert calls == want, (calls, want)  # embeddings, then per layer: attention probabilities, FFN output


def gen_full():
    """The real TVTS class (DistilBERT-base + ViT-B/16 with tubelets + sorting head, 170 M parameters) at B=2, 4 frames,
    mask 0.75 (the reference loader's clip shape, v1/configs/dist-yt-pt.json)."""
    ns = import_reference_v1()
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    m = ns.tvts.TVTS(args, video_params={"arch_config": "base_patch16_224", "num_frames": 16},
                     text_params={"model": "distilbert-base-uncased", "pretrained": True, "input": "text"},
                     projection="minimal", load_checkpoint="")
    a = V.ARCH
    names = list(m.state_dict().keys())
    assert names == list(V.param_shapes(a).keys()), [n for n in names if n not in V.param_shapes(a)][:5]
    m.load_state_dict(V.synth_params(a, seed=41), strict=True)
    batch = V.synth_batch(a, B=2, T=4, seed=42, caption_len=14)
    sel = {"g_conv": ("video_model.patch_embed.proj.weight", (slice(0, 2),)),
           "g_temporal": ("video_model.temporal_embed", (slice(None), slice(None), slice(0, 32))),
           "g_qkv11": ("video_model.blocks.11.attn.qkv.weight", (slice(0, 8), slice(0, 32))),
           "g_qlin5": ("text_model.transformer.layer.5.attention.q_lin.weight", (slice(0, 8), slice(0, 32))),
           "g_txtproj": ("txt_proj.1.weight", (slice(0, 8), slice(0, 32))),
           "g_vidproj": ("vid_proj.0.weight", (slice(0, 8), slice(0, 32))),
           "g_head": ("pred_model.head.weight", (slice(None), slice(0, 64)))}
    _run(m, batch, ns, sel, "v1_full", dict(seed=41, batch_seed=42, B=2, T=4, caption_len=14))


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "dropout", "full"]
    torch.manual_seed(0)
    if "tiny" in which:
        gen_tiny()
    if "dropout" in which:
        gen_tiny_dropout()
    if "full" in which:
        gen_full()
