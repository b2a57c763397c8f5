#!/usr/bin/env python3
"""Generate the parity fixtures in tests/golden/ by RUNNING THE REFERENCE (container only).

The reference (/root/reference, TencentARC/TVTS v2) is imported read-only with import shims for
third-party modules that are absent from this image (timm, torchvision, humanize, ftfy ...).
No reference file is edited or copied: this script only calls into it and stores *data*
(inputs, expected outputs, expected gradients) as small .npz files.  The reference never
travels to the GPU box; the fixtures do.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py block vit  # a subset

Synthetic parameters and batches come from oracle/tvts_oracle.py (synth_params / synth_batch),
so the test side can regenerate identical inputs without storing 186 M parameters.
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
REF = os.environ.get("TVTS_REFERENCE", "/root/reference/v2")
sys.path.insert(0, ROOT)

from oracle import tvts_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- import shims
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    return m


def import_reference():
    """Returns a namespace of the reference classes on the hot path."""
    if "ref_ns" in globals():
        return globals()["ref_ns"]
    ident = lambda *a, **k: None  # noqa: E731
    _stub("timm"); _stub("timm.models")
    _stub("timm.models.layers", StdConv2dSame=object, DropPath=torch.nn.Identity, to_2tuple=ident,
          trunc_normal_=ident)
    _stub("torchvision"); _stub("torchvision.utils", make_grid=ident)
    _stub("torchvision.ops"); _stub("torchvision.ops.misc", FrozenBatchNorm2d=torch.nn.Identity)
    _stub("humanize")
    # `base`: only BaseModel is needed for the model classes
    base = _stub("base")
    bm = _load("base.base_model", os.path.join(REF, "base/base_model.py"))
    base.BaseModel = bm.BaseModel
    # `utils.util`
    utils = _stub("utils")
    uu = _load("utils.util", os.path.join(REF, "utils/util.py"))
    utils.util = uu
    utils.inf_loop = uu.inf_loop
    # `CLIP.clip` : real model.py, stubbed loader (no pretrained weights in the tree)
    clip_pkg = _stub("CLIP")
    clip_model = _load("CLIP.clip.model", os.path.join(REF, "CLIP/clip/model.py"))
    clip_mod = _stub("CLIP.clip", model=clip_model)
    clip_pkg.clip = clip_mod

    def fake_load(path, device="cpu", **kw):
        patch = 32 if "32" in path else 16
        torch.manual_seed(1234)
        return clip_model.CLIP(512, 224, 12, 768, patch, 77, 49408, 512, 8, 12), None
    clip_mod.load = fake_load
    model_pkg = _stub("model")
    st = _load("model.sort_transformer", os.path.join(REF, "model/sort_transformer.py"))
    ve16 = _load("model.video_encoder_ViT_B_16", os.path.join(REF, "model/video_encoder_ViT_B_16.py"))
    ve32 = _load("model.video_encoder_ViT_B_32", os.path.join(REF, "model/video_encoder_ViT_B_32.py"))
    loss = _load("model.loss", os.path.join(REF, "model/loss.py"))
    m32 = _load("model.model_dist_TVTSv2_ViT_B_32", os.path.join(REF, "model/model_dist_TVTSv2_ViT_B_32.py"))
    m16 = _load("model.model_dist_TVTSv2_ViT_B_16", os.path.join(REF, "model/model_dist_TVTSv2_ViT_B_16.py"))
    model_pkg.sort_transformer = st
    ns = types.SimpleNamespace(clip_model=clip_model, sort=st, ve16=ve16, ve32=ve32, loss=loss, m32=m32, m16=m16)
    globals()["ref_ns"] = ns
    return ns


def import_reference_allgather():
    """trainer.trainer.AllGather_multi (needs `logger` + base_trainer shims)."""
    ns = import_reference()
    _stub("logger", TensorboardWriter=object)
    bt = _load("base.base_trainer", os.path.join(REF, "base/base_trainer.py"))
    sys.modules["base"].Multi_BaseTrainer_dist = bt.Multi_BaseTrainer_dist
    sys.modules["base"].BaseTrainer = bt.BaseTrainer
    tr = _load("trainer.trainer", os.path.join(REF, "trainer/trainer.py"))
    ns.trainer = tr
    return tr


# ----------------------------------------------------------------------------- helpers
def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def sub(P, prefix):
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


class TinyRefModel(torch.nn.Module):
    """Reference sub-modules at the tiny architecture, wired by the reference's own
    TVTSv2_B_32.forward / compute_text / compute_video (called as unbound functions)."""

    def __init__(self, ns, arch, P):
        super().__init__()
        a = arch
        clip = ns.clip_model.CLIP(a["embed"], a["image"], 1, a["width"], a["patch"], a["context"], a["vocab"],
                                  a["text_width"], a["text_heads"], a["text_layers"])
        self.text_model = clip.transformer
        self.text_token_embedding = clip.token_embedding
        self.text_positional_embedding = clip.positional_embedding
        self.text_ln_final = clip.ln_final
        self.text_projection = clip.text_projection
        self.video_model = ns.ve16.VisionTransformer(input_resolution=a["image"], patch_size=a["patch"],
                                                     width=a["width"], layers=a["layers"], heads=a["heads"],
                                                     output_dim=a["embed"], num_frames=a["num_frames"],
                                                     mask_ratio=a["mask_ratio"])
        self.pred_model = ns.sort.SortTransformer(num_classes=a["n_trans"], embed_dim=a["embed"],
                                                  num_heads=a["sort_heads"])
        missing = self.load_state_dict(P, strict=True)
        self._ref = ns.m32.TVTSv2_B_32

    def compute_text(self, t):
        return self._ref.compute_text(self, t)

    def compute_video(self, v, k):
        return self._ref.compute_video(self, v, k)

    def forward(self, data, return_embeds=True):
        return self._ref.forward(self, data, return_embeds)


def ref_losses(ns, te, ve, pred, label):
    loss1 = ns.loss.NormSoftmaxLoss()(ns.m32.sim_matrix(ve, te))
    loss2 = torch.nn.CrossEntropyLoss()(pred.reshape(-1, pred.shape[-1]), label.reshape(-1)) * 2
    return loss1, loss2


# ----------------------------------------------------------------------------- fixtures
def gen_block():
    """One ResidualSpaceTimeAttentionBlock (video_encoder_ViT_B_16.py:94-124), fwd + grads."""
    ns = import_reference()
    arch = O.tiny_arch()
    P = O.synth_params(arch, seed=11)
    pre = "video_model.transformer.resblocks.1."
    blk = ns.ve16.ResidualSpaceTimeAttentionBlock(arch["width"], arch["heads"])
    blk.load_state_dict(sub(P, pre), strict=True)
    B, T, n = 2, 3, 5
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 1 + T * n, arch["width"], generator=g, requires_grad=True)
    r = torch.randn(B, 1 + T * n, arch["width"], generator=g)
    y = blk(x, "b (f n) d", "(b f) n d", "b (f n) d", "(b n) f d", n, T)
    (y * r).sum().backward()
    grads = {"g_" + k: v.grad for k, v in blk.named_parameters()}
    save("block_tiny", x=x, r=r, y=y, gx=x.grad, T=T, n=n, seed=11, **grads)


def gen_vit():
    """VisionTransformer.forward with tube masking (video_encoder_ViT_B_16.py:176-235)."""
    ns = import_reference()
    arch = O.tiny_arch()
    P = O.synth_params(arch, seed=12)
    vit = ns.ve16.VisionTransformer(arch["image"], arch["patch"], arch["width"], arch["layers"], arch["heads"],
                                    arch["embed"], num_frames=arch["num_frames"], mask_ratio=arch["mask_ratio"])
    vit.load_state_dict(sub(P, "video_model."), strict=True)
    batch = O.synth_batch(arch, B=2, T=3, seed=3)
    out = vit(batch["video"], batch["keep_ind"])
    g = torch.Generator().manual_seed(6)
    r = torch.randn(out.shape, generator=g)
    (out * r).sum().backward()
    sel = ["conv1.weight", "positional_embedding", "temporal_embedding", "class_embedding", "proj",
           "ln_pre.weight", "transformer.resblocks.0.timeattn.qkv.weight", "transformer.resblocks.1.mlp.c_fc.bias"]
    grads = {"g_" + k: dict(vit.named_parameters())[k].grad for k in sel}
    save("vit_tiny", out=out, r=r, seed=12, batch_seed=3, B=2, T=3, **grads)


def gen_text():
    """compute_text (model_dist_TVTSv2_ViT_B_32.py:97-111) on the CLIP text tower."""
    ns = import_reference()
    arch = O.tiny_arch()
    P = O.synth_params(arch, seed=13)
    m = TinyRefModel(ns, arch, P)
    batch = O.synth_batch(arch, B=3, T=1, seed=4, caption_len=9)
    ids = batch["text"].clone()
    ids[1, 5] = arch["vocab"] - 1; ids[1, 6:] = 0  # a ragged row: EOT earlier than the others
    before, emb = m.compute_text(ids)
    g = torch.Generator().manual_seed(7)
    r = torch.randn(emb.shape, generator=g)
    (emb * r).sum().backward()
    save("text_tiny", ids=ids, emb=emb, r=r, seed=13,
         g_tok=m.text_token_embedding.weight.grad, g_pos=m.text_positional_embedding.grad,
         g_proj=m.text_projection.grad, g_qkv0=m.text_model.resblocks[0].attn.in_proj_weight.grad,
         g_lnf=m.text_ln_final.weight.grad)


def gen_sort():
    """SortTransformer.forward (sort_transformer.py:124-142)."""
    ns = import_reference()
    arch = O.tiny_arch()
    P = O.synth_params(arch, seed=14)
    sh = ns.sort.SortTransformer(num_classes=4, embed_dim=arch["embed"], num_heads=arch["sort_heads"])
    sh.load_state_dict(sub(P, "pred_model."), strict=True)
    g = torch.Generator().manual_seed(8)
    text = torch.randn(2, 4, arch["embed"], generator=g)
    tok = torch.randn(2, 11, arch["embed"], generator=g, requires_grad=True)
    out = sh(text, tok)
    r = torch.randn(out.shape, generator=g)
    (out * r).sum().backward()
    save("sort_tiny", text=text, tok=tok, out=out, r=r, g_tok=tok.grad, seed=14,
         g_type=sh.type_embed.grad, g_head=sh.head.weight.grad, g_qkv1=sh.blocks[1].attn.qkv.weight.grad,
         g_fc1b=sh.blocks[0].mlp.fc1.bias.grad)


def gen_model_tiny():
    """Whole TVTSv2 forward (reference's own forward wiring) + both losses + all grads, tiny arch."""
    ns = import_reference()
    arch = O.tiny_arch()
    P = O.synth_params(arch, seed=15)
    m = TinyRefModel(ns, arch, P)
    batch = O.synth_batch(arch, B=3, T=3, seed=9, caption_len=10)
    te, ve, pred = m(batch)
    loss1, loss2 = ref_losses(ns, te, ve, pred, batch["label"])
    (loss1 + loss2).backward()
    gn = {k: v.grad.norm() for k, v in m.named_parameters() if v.grad is not None}
    total = torch.sqrt(sum(v ** 2 for v in gn.values()))
    sel = ["video_model.proj", "video_model.transformer.resblocks.0.timeattn.proj.weight",
           "text_projection", "text_model.resblocks.2.mlp.c_proj.weight", "pred_model.head.weight",
           "video_model.conv1.weight", "text_positional_embedding"]
    grads = {"g_" + k: dict(m.named_parameters())[k].grad for k in sel}
    save("model_tiny", te=te, ve=ve, pred=pred, loss1=loss1, loss2=loss2, grad_norm=total, seed=15, batch_seed=9,
         B=3, T=3, caption_len=10, gn_names=np.array(list(gn.keys())), gn_vals=torch.stack(list(gn.values())),
         **grads)
    # WebVid-style batch: NT = 1, no sort head (model_dist..:87-90, trainer.py:494)
    m.zero_grad()
    b1 = O.synth_batch(arch, B=3, T=3, seed=10, n_trans=1, caption_len=10)
    te, ve, pred = m(b1)
    assert pred is None
    l1 = ns.loss.NormSoftmaxLoss()(ns.m32.sim_matrix(ve, te))
    l1.backward()
    save("model_tiny_nt1", te=te, ve=ve, loss1=l1, seed=15, batch_seed=10,
         g_proj=m.video_model.proj.grad, g_tproj=m.text_projection.grad)


def gen_model_b32():
    """The real TVTSv2_B_32 class at BASELINE config 1 (B=2, T=4, 49 patches, context 77)."""
    ns = import_reference()
    arch = O.ARCHS["B_32"]
    P = O.synth_params(arch, seed=0)
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    m = ns.m32.TVTSv2_B_32(args, load_checkpoint="")
    m.load_state_dict(P, strict=True)
    assert list(m.state_dict().keys()) == list(P.keys()), "state-dict key order differs from param_shapes()"
    batch = O.synth_batch(arch, B=2, T=4, seed=0)
    te, ve, pred = m(batch)
    loss1, loss2 = ref_losses(ns, te, ve, pred, batch["label"])
    (loss1 + loss2).backward()
    names, vals = [], []
    for k, v in m.named_parameters():
        if v.grad is not None:
            names.append(k); vals.append(v.grad.norm())
    total = torch.sqrt(sum(v ** 2 for v in vals))
    pd = dict(m.named_parameters())
    sel = {"g_video_proj": pd["video_model.proj"].grad[:8, :16],
           "g_text_proj": pd["text_projection"].grad[:8, :16],
           "g_head": pd["pred_model.head.weight"].grad,
           "g_cfc11": pd["video_model.transformer.resblocks.11.mlp.c_fc.weight"].grad[:8, :16],
           "g_tqkv0": pd["video_model.transformer.resblocks.0.timeattn.qkv.weight"].grad[:8, :16],
           "g_temporal": pd["video_model.temporal_embedding"].grad[:, :16],
           "g_textqkv10": pd["text_model.resblocks.10.attn.in_proj_weight"].grad[:8, :16]}
    save("model_b32_cfg1", te=te, ve=ve, pred=pred, loss1=loss1, loss2=loss2, grad_norm=total, seed=0,
         batch_seed=0, B=2, T=4, gn_names=np.array(names), gn_vals=torch.stack(vals), **sel)


def import_reference_h14():
    """H/14 classes: video_encoder_ViT_H_14 + OpenCLIP/transformer.py + model_dist_TVTSv2_ViT_H_14 (the OpenCLIP
    package __init__ pulls tokenizers/pretrained tables that are irrelevant here, so only its transformer/utils load)."""
    ns = import_reference()
    if hasattr(ns, "mh"):
        return ns
    oc = _stub("OpenCLIP")
    oc.utils = _load("OpenCLIP.utils", os.path.join(REF, "OpenCLIP/utils.py"))
    oc.transformer = _load("OpenCLIP.transformer", os.path.join(REF, "OpenCLIP/transformer.py"))
    ns.oc = oc.transformer
    ns.veh = _load("model.video_encoder_ViT_H_14", os.path.join(REF, "model/video_encoder_ViT_H_14.py"))
    ns.mh = _load("model.model_dist_TVTSv2_ViT_H_14", os.path.join(REF, "model/model_dist_TVTSv2_ViT_H_14.py"))
    return ns


def h_tiny_arch():
    return O.tiny_arch(name="H_14", image=28, patch=7, width=80, heads=2, layers=2, embed=32, text_width=32,
                       text_heads=2, text_layers=3, text_tune_from=1, act="gelu", tail="pooled_and_patches",
                       block_order="openclip", mask_ratio=0.7)


class TinyRefModelH(torch.nn.Module):
    """Reference H/14 sub-modules at a tiny size, wired by TVTSv2_H_14.forward / compute_text / compute_video."""

    def __init__(self, ns, arch, P):
        super().__init__()
        a = arch
        self.text_model = ns.oc.Transformer(width=a["text_width"], layers=a["text_layers"], heads=a["text_heads"],
                                            act_layer=torch.nn.GELU, norm_layer=ns.oc.LayerNorm)
        self.text_token_embedding = torch.nn.Embedding(a["vocab"], a["text_width"])
        self.text_positional_embedding = torch.nn.Parameter(torch.empty(a["context"], a["text_width"]))
        self.text_ln_final = ns.oc.LayerNorm(a["text_width"])
        self.text_projection = torch.nn.Parameter(torch.empty(a["text_width"], a["embed"]))
        # OpenCLIP/transformer.py:694-700 / model.py build_attention_mask: additive causal mask
        self.text_attn_mask = torch.full((a["context"], a["context"]), float("-inf")).triu_(1)
        self.video_model = ns.veh.VisionTransformer(
            image_size=a["image"], patch_size=a["patch"], width=a["width"], layers=a["layers"], heads=a["heads"],
            mlp_ratio=4.0, output_dim=a["embed"], act_layer=torch.nn.GELU, norm_layer=ns.veh.LayerNorm,
            num_frames=a["num_frames"], mask_ratio=a["mask_ratio"])
        self.pred_model = ns.sort.SortTransformer(num_classes=a["n_trans"], embed_dim=a["embed"],
                                                  num_heads=a["sort_heads"])
        self._ref = ns.mh.TVTSv2_H_14

    def compute_text(self, t):
        return self._ref.compute_text(self, t)

    def compute_video(self, v, k):
        return self._ref.compute_video(self, v, k)

    def forward(self, data, return_embeds=True):
        return self._ref.forward(self, data, return_embeds)


def gen_model_h_tiny():
    """H/14 structure: OpenCLIP text tower (GELU, un-truncated causal context), ln_post on CLS only, patch tokens
    without CLS to the sort head, state-dict registration order."""
    ns = import_reference_h14()
    arch = h_tiny_arch()
    m = TinyRefModelH(ns, arch, None)
    names = list(m.state_dict().keys())
    assert names == list(O.param_shapes(arch).keys()), [(a, b) for a, b in zip(names, O.param_shapes(arch)) if a != b][:5]
    P = O.synth_params(arch, seed=21)
    m.load_state_dict(P, strict=True)
    batch = O.synth_batch(arch, B=3, T=3, seed=12, caption_len=10)
    te, ve, pred = m(batch)
    loss1, loss2 = ref_losses(ns, te, ve, pred, batch["label"])
    (loss1 + loss2).backward()
    gn = {k: v.grad.norm() for k, v in m.named_parameters() if v.grad is not None}
    total = torch.sqrt(sum(v ** 2 for v in gn.values()))
    sel = ["video_model.proj", "video_model.ln_post.weight", "video_model.transformer.resblocks.1.attn.qkv.weight",
           "video_model.conv1.weight", "text_projection", "text_model.resblocks.2.mlp.c_fc.weight",
           "pred_model.type_embed"]
    grads = {"g_" + k: dict(m.named_parameters())[k].grad for k in sel}
    save("model_h_tiny", te=te, ve=ve, pred=pred, loss1=loss1, loss2=loss2, grad_norm=total, seed=21, batch_seed=12,
         B=3, T=3, caption_len=10, names=np.array(names), gn_names=np.array(list(gn.keys())),
         gn_vals=torch.stack(list(gn.values())), **grads)


def gen_model_h14():
    """The real TVTSv2_H_14 class at full size (ViT-H/14 + 24-layer OpenCLIP text tower, 1.22 G parameters), B=2, T=4.
    OpenCLIP.create_model (pretrained weights, not in the tree) is replaced by a factory that builds the reference's own
    OpenCLIP TextTransformer from the reference's ViT-H-14.json; every parameter is then overwritten by synth_params."""
    import json
    ns = import_reference_h14()
    with open(os.path.join(REF, "OpenCLIP/model_configs/ViT-H-14.json")) as f:
        cfg = json.load(f)

    def create_model(name, pretrained=None, cache_dir=None, **kw):
        assert name == "ViT-H-14"
        t = cfg["text_cfg"]
        text = ns.oc.TextTransformer(context_length=t["context_length"], vocab_size=t["vocab_size"], width=t["width"],
                                     heads=t["heads"], layers=t["layers"], output_dim=cfg["embed_dim"],
                                     act_layer=torch.nn.GELU, norm_layer=ns.oc.LayerNorm)
        visual = types.SimpleNamespace(state_dict=lambda: {})
        return types.SimpleNamespace(transformer=text.transformer, token_embedding=text.token_embedding,
                                     positional_embedding=text.positional_embedding, ln_final=text.ln_final,
                                     text_projection=text.text_projection, attn_mask=text.attn_mask, visual=visual)
    sys.modules["OpenCLIP"].create_model = create_model
    arch = O.ARCHS["H_14"]
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    cwd = os.getcwd()
    os.chdir(REF)  # the class opens 'OpenCLIP/model_configs/ViT-H-14.json' relative to the v2/ directory
    try:
        m = ns.mh.TVTSv2_H_14(args, load_checkpoint="")
    finally:
        os.chdir(cwd)
    P = O.synth_params(arch, seed=0)
    assert list(m.state_dict().keys()) == list(P.keys()), "state-dict key order differs from param_shapes()"
    m.load_state_dict(P, strict=True)
    del P
    batch = O.synth_batch(arch, B=2, T=4, seed=0)
    te, ve, pred = m(batch)
    loss1, loss2 = ref_losses(ns, te, ve, pred, batch["label"])
    (loss1 + loss2).backward()
    names, vals = [], []
    for k, v in m.named_parameters():
        if v.grad is not None:
            names.append(k); vals.append(v.grad.norm())
    total = torch.sqrt(sum(v ** 2 for v in vals))
    pd = dict(m.named_parameters())
    sel = {"g_video_proj": pd["video_model.proj"].grad[:8, :16],
           "g_text_proj": pd["text_projection"].grad[:8, :16],
           "g_head": pd["pred_model.head.weight"].grad[:, :64],
           "g_conv": pd["video_model.conv1.weight"].grad[:4].reshape(4, -1),
           "g_lnpost": pd["video_model.ln_post.weight"].grad,
           "g_cfc31": pd["video_model.transformer.resblocks.31.mlp.c_fc.weight"].grad[:8, :16],
           "g_tqkv0": pd["video_model.transformer.resblocks.0.timeattn.qkv.weight"].grad[:8, :16],
           "g_temporal": pd["video_model.temporal_embedding"].grad[:, :16],
           "g_textqkv20": pd["text_model.resblocks.20.attn.in_proj_weight"].grad[:8, :16]}
    save("model_h14_cfg3", te=te, ve=ve, pred=pred, loss1=loss1, loss2=loss2, grad_norm=total, seed=0,
         batch_seed=0, B=2, T=4, gn_names=np.array(names), gn_vals=torch.stack(vals), **sel)


def gen_metrics():
    """v2/model/metric.py t2v_metrics / v2t_metrics on similarity matrices with and without ties, 1 and 2 captions
    per video (the functions are executed as they are; ipdb is the only import that needs a stub)."""
    _stub("ipdb")
    mm = _load("ref_metric", os.path.join(REF, "model/metric.py"))
    rng = np.random.RandomState(5)
    out = {}
    cases = {"rand_square": rng.randn(37, 37).astype(np.float32),
             "ties_square": np.round(rng.randn(29, 29) * 1.5).astype(np.float32) / 2,   # many exact ties
             "two_caps": rng.randn(40, 20).astype(np.float32),
             "two_caps_ties": np.round(rng.randn(24, 12) * 2).astype(np.float32)}
    cases["rand_square"][np.arange(37), np.arange(37)] += 1.5  # a model that has learnt something
    keys = ["R1", "R5", "R10", "R50", "MedR", "MeanR", "geometric_mean_R1-R5-R10"]
    for name, sims in cases.items():
        out["sims_" + name] = sims
        for fn in ("t2v_metrics", "v2t_metrics"):
            res = getattr(mm, fn)(sims.copy())
            out[f"{fn}_{name}"] = np.array([float(res[k]) for k in keys], dtype=np.float64)
    # MSRVTT-style missing captions (query_masks): np.bool, which metric.py:107 still spells, left numpy in 1.24
    if not hasattr(np, "bool"):
        np.bool = bool
    mask = np.ones((20, 2), dtype=np.float32)
    mask[3, 1] = mask[11, 1] = mask[17, 1] = 0
    out["query_mask_two_caps"] = mask
    for fn in ("t2v_metrics", "v2t_metrics"):
        res = getattr(mm, fn)(cases["two_caps"].copy(), query_masks=mask.copy())
        out[f"{fn}_two_caps_masked"] = np.array([float(res[k]) for k in keys], dtype=np.float64)
    save("metrics", keys=np.array(keys), **out)


def downstream_arch(name):
    return dict(O.ARCHS[name], mask_ratio=0.0, sort_head=False)


def gen_downstream():
    """The reference's inference-only TVTSv2_B_16 (v2/downstream/model_TVTSv2_ViT_B_16.py): no tube masking (196 patches,
    keep_ind broadcast over the batch), no sort head; a retrieval-style batch (B=2, T=4, one caption each) and the
    zero-shot scripts' text pass beside dummy single frames."""
    ns = import_reference()
    _stub("downstream")
    dm = _load("downstream.model_TVTSv2_ViT_B_16", os.path.join(REF, "downstream/model_TVTSv2_ViT_B_16.py"))
    arch = downstream_arch("B_16")
    m = dm.TVTSv2_B_16(load_checkpoint="")
    P = O.synth_params(arch, seed=0)
    assert list(m.state_dict().keys()) == list(P.keys()), "downstream state-dict keys differ from param_shapes(sort_head=False)"
    m.load_state_dict(P, strict=True)
    m.eval()
    b = O.synth_batch(O.ARCHS["B_16"], B=2, T=4, seed=3, n_trans=1)
    keep = torch.arange(196).unsqueeze(0)
    with torch.no_grad():
        keep_b = keep.expand(2, -1)  # the datasets deliver one index row per sample
        te, ve = m({"text": b["text"], "video": b["video"], "keep_ind": keep_b}, return_embeds=True)
        sims = m({"text": b["text"], "video": b["video"], "keep_ind": keep_b}, return_embeds=False)
        g = torch.Generator().manual_seed(4)
        prompts = O.synth_batch(O.ARCHS["B_16"], B=3, T=1, seed=5, n_trans=1, caption_len=12)["text"]
        cls_emb, _ = m({"text": prompts, "video": torch.zeros(3, 3, 224, 224), "keep_ind": keep}, return_embeds=True)
    save("downstream_b16", te=te, ve=ve, sims=sims, prompts=prompts, cls_emb=cls_emb, seed=0, batch_seed=3)


def gen_transform():
    """The reference's eval transform tail on uint8 frames (video_transforms/videoaug.py:20-27 after the cv2 resize):
    CenterCrop -> ClipToTensor -> Normalize, executed from video_transform.py / functional.py (cv2, skimage and
    torchvision are import-time dependencies of those files only; stubbed)."""
    _stub("cv2"); _stub("skimage"); _stub("skimage.transform"); _stub("torchvision")
    _stub("torchvision.transforms")
    vt_pkg = _stub("video_transforms")
    vt_pkg.functional = _load("video_transforms.functional", os.path.join(REF, "video_transforms/functional.py"))
    vt = _load("video_transforms.video_transform", os.path.join(REF, "video_transforms/video_transform.py"))
    rng = np.random.RandomState(11)
    frames = rng.randint(0, 256, size=(3, 40, 44, 3)).astype(np.uint8)  # T x H0 x W0 x 3, as TensorToNumpy + Resize hand over
    clip = [frames[t] for t in range(frames.shape[0])]
    clip = vt.CenterCrop(32)(clip)
    ten = vt.ClipToTensor(channel_nb=3)(clip)                       # C x T x H x W
    ten = vt.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(ten)
    out = ten.permute(1, 0, 2, 3)                                   # base_dataset.py:126 -> T x C x H x W
    save("transform", frames=frames, out=out, image=32)


def gen_transform_resize():
    """The reference's WHOLE eval transform on decoder-sized pictures (video_transforms/videoaug.py:19-27): PIL images as
    TensorToNumpy leaves them -> Resize(int(crop * 1.2)) (nearest, the class default) -> CenterCrop -> ClipToTensor -> Normalize,
    and the train chain's RandomCrop position replaced by a fixed offset (the crop op itself, functional.crop_clip)."""
    from PIL import Image
    _stub("cv2"); _stub("skimage"); _stub("skimage.transform"); _stub("torchvision")
    _stub("torchvision.transforms")
    vt_pkg = _stub("video_transforms")
    vt_pkg.functional = _load("video_transforms.functional", os.path.join(REF, "video_transforms/functional.py"))
    vt = _load("video_transforms.video_transform", os.path.join(REF, "video_transforms/video_transform.py"))
    rng = np.random.RandomState(12)
    out = {}
    for tag, (hs, ws) in {"wide": (57, 90), "tall": (101, 64), "same": (48, 70)}.items():
        frames = rng.randint(0, 256, size=(3, hs, ws, 3)).astype(np.uint8)
        clip = [Image.fromarray(frames[t]).convert("RGB") for t in range(3)]
        clip = vt.Resize(int(40 * 1.2))(clip)
        ten = vt.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(vt.ClipToTensor(channel_nb=3)(vt.CenterCrop(40)(clip)))
        out["frames_" + tag] = frames
        out["out_" + tag] = ten.permute(1, 0, 2, 3)
        out["resized_hw_" + tag] = np.array([clip[0].size[1], clip[0].size[0]])
        # a fixed off-centre crop (what RandomCrop does once its offsets are drawn, video_transform.py:204-232)
        y1, x1 = 3, 5
        crop = vt_pkg.functional.crop_clip(clip, y1, x1, 40, 40)
        ten2 = vt.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(vt.ClipToTensor(channel_nb=3)(crop))
        out["out_crop_" + tag] = ten2.permute(1, 0, 2, 3)
    save("transform_resize", image=40, size=48, crop_yx=np.array([3, 5]), **out)


def gen_tokenize():
    """clip.tokenize (CLIP/clip/clip.py:197-237) on a few captions, with and without truncation: the rows the caption cache must
    reproduce.  ftfy is absent from the image: its fix_text is the identity on these ASCII captions."""
    _stub("ftfy", fix_text=lambda t: t)
    _stub("torchvision"); _stub("torchvision.transforms", Compose=object, Resize=object, CenterCrop=object, ToTensor=object,
                                Normalize=object, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    pkg = _stub("CLIP"); sub = _stub("CLIP.clip")
    _load("CLIP.clip.simple_tokenizer", os.path.join(REF, "CLIP/clip/simple_tokenizer.py"))
    _load("CLIP.clip.model", os.path.join(REF, "CLIP/clip/model.py"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("CLIP.clip.clip", os.path.join(REF, "CLIP/clip/clip.py"),
                                                  submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "CLIP.clip"
    sys.modules["CLIP.clip.clip"] = mod
    spec.loader.exec_module(mod)
    caps = ["a person is cooking pasta in a kitchen", "then add the onions and stir", "now we cut the wood",
            "a person is cooking pasta in a kitchen", "hello world", " ".join(["very long caption word"] * 30)]
    toks = mod.tokenize(caps, truncate=True)
    save("tokenize", captions=np.array(caps), tokens=toks.numpy().astype(np.int32),
         tokens_short=mod.tokenize(caps[:5]).numpy().astype(np.int32))


def gen_downstream_more():
    """The other two downstream copies: TVTSv2_B_32 (49 unmasked patches: the fused SPACE kernels' range) and the full-size
    TVTSv2_H_14 (256 unmasked patches, head dim 80, pooled tail; OpenCLIP.create_model replaced as in gen_model_h14)."""
    import json
    ns = import_reference_h14()
    if "downstream" not in sys.modules:
        _stub("downstream")
    d32 = _load("downstream.model_TVTSv2_ViT_B_32", os.path.join(REF, "downstream/model_TVTSv2_ViT_B_32.py"))
    arch = downstream_arch("B_32")
    m = d32.TVTSv2_B_32(load_checkpoint="")
    P = O.synth_params(arch, seed=0)
    assert list(m.state_dict().keys()) == list(P.keys())
    m.load_state_dict(P, strict=True); m.eval()
    b = O.synth_batch(O.ARCHS["B_32"], B=2, T=5, seed=6, n_trans=1)
    keep = torch.arange(49).unsqueeze(0).expand(2, -1)
    with torch.no_grad():
        te, ve = m({"text": b["text"], "video": b["video"], "keep_ind": keep}, return_embeds=True)
    save("downstream_b32", te=te, ve=ve, seed=0, batch_seed=6)
    del m, P
    with open(os.path.join(REF, "OpenCLIP/model_configs/ViT-H-14.json")) as f:
        cfg = json.load(f)

    def create_model(name, pretrained=None, cache_dir=None, **kw):
        t = cfg["text_cfg"]
        text = ns.oc.TextTransformer(context_length=t["context_length"], vocab_size=t["vocab_size"], width=t["width"],
                                     heads=t["heads"], layers=t["layers"], output_dim=cfg["embed_dim"],
                                     act_layer=torch.nn.GELU, norm_layer=ns.oc.LayerNorm)
        return types.SimpleNamespace(transformer=text.transformer, token_embedding=text.token_embedding,
                                     positional_embedding=text.positional_embedding, ln_final=text.ln_final,
                                     text_projection=text.text_projection, attn_mask=text.attn_mask,
                                     visual=types.SimpleNamespace(state_dict=lambda: {}))
    sys.modules["OpenCLIP"].create_model = create_model
    dh = _load("downstream.model_TVTSv2_ViT_H_14", os.path.join(REF, "downstream/model_TVTSv2_ViT_H_14.py"))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        m = dh.TVTSv2_H_14(load_checkpoint="")
    finally:
        os.chdir(cwd)
    arch = downstream_arch("H_14")
    P = O.synth_params(arch, seed=0)
    assert list(m.state_dict().keys()) == list(P.keys())
    m.load_state_dict(P, strict=True); m.eval()
    del P
    b = O.synth_batch(O.ARCHS["H_14"], B=1, T=2, seed=7, n_trans=1)
    with torch.no_grad():
        te, ve = m({"text": b["text"], "video": b["video"], "keep_ind": torch.arange(256).unsqueeze(0)}, return_embeds=True)
    save("downstream_h14", te=te, ve=ve, seed=0, batch_seed=7)


def gen_model_b16():
    """The real TVTSv2_B_16 class (the headline architecture: 196 patches, tube mask 0.5 -> 98 kept) at B=2, T=4."""
    ns = import_reference()
    arch = O.ARCHS["B_16"]
    P = O.synth_params(arch, seed=0)
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    m = ns.m16.TVTSv2_B_16(args, load_checkpoint="")
    m.load_state_dict(P, strict=True)
    assert list(m.state_dict().keys()) == list(P.keys())
    batch = O.synth_batch(arch, B=2, T=4, seed=1)
    te, ve, pred = m(batch)
    loss1, loss2 = ref_losses(ns, te, ve, pred, batch["label"])
    (loss1 + loss2).backward()
    names, vals = [], []
    for k, v in m.named_parameters():
        if v.grad is not None:
            names.append(k); vals.append(v.grad.norm())
    total = torch.sqrt(sum(v ** 2 for v in vals))
    pd = dict(m.named_parameters())
    sel = {"g_video_proj": pd["video_model.proj"].grad[:8, :16],
           "g_head": pd["pred_model.head.weight"].grad,
           "g_conv": pd["video_model.conv1.weight"].grad[:4].reshape(4, -1),
           "g_cfc11": pd["video_model.transformer.resblocks.11.mlp.c_fc.weight"].grad[:8, :16],
           "g_tqkv0": pd["video_model.transformer.resblocks.0.timeattn.qkv.weight"].grad[:8, :16],
           "g_pos": pd["video_model.positional_embedding"].grad[:, :16],
           "g_textqkv10": pd["text_model.resblocks.10.attn.in_proj_weight"].grad[:8, :16]}
    save("model_b16_cfg2", te=te, ve=ve, pred=pred, loss1=loss1, loss2=loss2, grad_norm=total, seed=0,
         batch_seed=1, B=2, T=4, gn_names=np.array(names), gn_vals=torch.stack(vals), **sel)


def _ctor_fixture(name, model, clip_sd, consumed_keys):
    """per-tensor record of a freshly constructed reference model: CRC-32 of the bytes, and what kind of value it holds"""
    sd = model.state_dict()
    names, crcs, kinds, means, stds = [], [], [], [], []
    by_crc = {O.tensor_crc(v): k for k, v in clip_sd.items()}
    for k, v in sd.items():
        c = O.tensor_crc(v)
        names.append(k); crcs.append(c)
        v = v.float()
        const = bool((v == v.flatten()[0]).all())
        kinds.append(0 if c in by_crc else (1 if const else 2))  # 0 = a pretrained tensor, bit for bit; 1 = constant; 2 = random
        means.append(float(v.mean())); stds.append(float(v.std()) if v.numel() > 1 else 0.0)
    save(name, names=np.array(names), crc=np.array(crcs, dtype=np.uint64), kind=np.array(kinds), mean=np.array(means),
         std=np.array(stds), const_value=np.array([float(sd[k].flatten()[0]) for k in names]),
         clip_keys=np.array(list(clip_sd.keys())), clip_shapes=np.array([",".join(str(int(x)) for x in v.shape) for v in clip_sd.values()]),
         consumed=np.array(sorted(consumed_keys)), seed=77)


def gen_ctor_init():
    """A13: what the REAL constructors leave in the state dict when load_checkpoint is empty.  The pretrained model is the
    reference's own CLIP class (B/16, B/32) / OpenCLIP towers (H/14) holding O.synth_clip_state_dict(<its own state-dict layout>, 77)
    -- the stand-in for CLIP/models/ViT-B-16.pt, which is not in the tree."""
    ns = import_reference()
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    for tag, cls_mod, patch in (("b16", ns.m16.TVTSv2_B_16, 16), ("b32", ns.m32.TVTSv2_B_32, 32)):
        holder = {}

        def fake_load(path, device="cpu", _patch=patch, _holder=holder, **kw):
            assert str(_patch) in path, path
            m = ns.clip_model.CLIP(512, 224, 12, 768, _patch, 77, 49408, 512, 8, 12)
            layout = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            sd = O.synth_clip_state_dict(layout, 77)
            m.load_state_dict(sd, strict=True)
            _holder["sd"] = sd
            return m, None
        sys.modules["CLIP.clip"].load = fake_load
        model = cls_mod(args, load_checkpoint="")
        sd = holder["sd"]
        consumed = [k for k in sd if not k.startswith("visual.")] + [k for k in sd if k.startswith("visual.")]
        _ctor_fixture("ctor_init_" + tag, model, sd, consumed)
        del model
    # H/14: OpenCLIP.create_model replaced by a factory that builds the reference's own OpenCLIP towers from ViT-H-14.json
    import json
    ns = import_reference_h14()
    with open(os.path.join(REF, "OpenCLIP/model_configs/ViT-H-14.json")) as f:
        cfg = json.load(f)
    holder = {}

    def create_model(name, pretrained=None, cache_dir=None, **kw):
        assert name == "ViT-H-14" and pretrained == "laion2b_s32b_b79k" and cache_dir == "OpenCLIP/models"
        t, v = cfg["text_cfg"], cfg["vision_cfg"]
        text = ns.oc.TextTransformer(context_length=t["context_length"], vocab_size=t["vocab_size"], width=t["width"],
                                     heads=t["heads"], layers=t["layers"], output_dim=cfg["embed_dim"],
                                     act_layer=torch.nn.GELU, norm_layer=ns.oc.LayerNorm)
        visual = ns.oc.VisionTransformer(image_size=v["image_size"], patch_size=v["patch_size"], width=v["width"], layers=v["layers"],
                                         heads=v["width"] // v["head_width"], mlp_ratio=4.0, output_dim=cfg["embed_dim"],
                                         act_layer=torch.nn.GELU, norm_layer=ns.oc.LayerNorm)
        layout = {k: tuple(x.shape) for k, x in text.state_dict().items() if k != "attn_mask"}
        layout.update({"visual." + k: tuple(x.shape) for k, x in visual.state_dict().items()})
        sd = O.synth_clip_state_dict(layout, 77)
        text.load_state_dict({k: x for k, x in sd.items() if not k.startswith("visual.")}, strict=False)
        visual.load_state_dict({k[7:]: x for k, x in sd.items() if k.startswith("visual.")}, strict=True)
        holder["sd"] = sd
        return types.SimpleNamespace(transformer=text.transformer, token_embedding=text.token_embedding,
                                     positional_embedding=text.positional_embedding, ln_final=text.ln_final,
                                     text_projection=text.text_projection, attn_mask=text.attn_mask, visual=visual)
    sys.modules["OpenCLIP"].create_model = create_model
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        model = ns.mh.TVTSv2_H_14(args, load_checkpoint="")
    finally:
        os.chdir(cwd)
    _ctor_fixture("ctor_init_h14", model, holder["sd"], list(holder["sd"]))


def gen_groups():
    """Name -> optimizer group, by executing the reference entrypoint's own grouping statements
    (train_dist_TVTSv2_ViT_B_16.py:66-107) on a module exposing the A13 parameter names."""
    src = open(os.path.join(REF, "train_dist_TVTSv2_ViT_B_16.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if "no_decay_names = [" in l)
    stop = next(i for i, l in enumerate(src) if "optimizer_grouped_parameters = [" in l)
    body = "\n".join(l[4:] if l.startswith("    ") else l for l in src[start:stop])
    arch = O.ARCHS["B_16"]

    class Fake:
        def named_parameters(self_inner):
            return [(k, torch.nn.Parameter(torch.zeros(1))) for k in O.param_shapes(arch)]
    env = {"model": Fake()}
    exec(compile(body, "<reference grouping>", "exec"), env)
    names, gid = [], []
    for gi, key in enumerate(["decay_new_params", "no_decay_new_params", "decay_clip_params", "no_decay_clip_params"]):
        for n, _ in env[key]:
            names.append(n); gid.append(gi)
    frozen = [k for k, p in env["model"].named_parameters()]
    grouped = set(names)
    frozen = [k for k in O.param_shapes(arch) if k not in grouped]
    save("param_groups_b16", names=np.array(names), group=np.array(gid), frozen=np.array(frozen))


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ns = import_reference()
    tr = import_reference_allgather()
    arch = O.tiny_arch()
    P = O.synth_params(arch, seed=16)
    m = TinyRefModel(ns, arch, P)
    ddp = torch.nn.parallel.DistributedDataParallel(m, find_unused_parameters=True)
    args = types.SimpleNamespace(local_rank=rank, rank=rank, world_size=world)
    batch = O.synth_batch(arch, B=2, T=2, seed=100 + rank, caption_len=8)
    te, ve, pred = ddp(batch)
    ve_all = tr.AllGather_multi.apply(ve, world, args)     # trainer.py:481-482
    te_all = tr.AllGather_multi.apply(te, world, args)
    loss1 = ns.loss.NormSoftmaxLoss()(ns.m32.sim_matrix(ve_all, te_all))
    loss2 = torch.nn.CrossEntropyLoss()(pred.reshape(-1, 4), batch["label"].reshape(-1)) * 2
    (loss1 + loss2).backward()
    pd = dict(m.named_parameters())
    sel = ["video_model.proj", "text_projection", "pred_model.head.weight",
           "video_model.transformer.resblocks.1.mlp.c_fc.weight", "text_token_embedding.weight"]
    gn = torch.sqrt(sum(p.grad.norm() ** 2 for p in m.parameters() if p.grad is not None))
    q.put((rank, float(loss1), float(loss2), float(gn), {k: pd[k].grad.clone().numpy() for k in sel}))
    dist.barrier()
    dist.destroy_process_group()


def gen_ddp2():
    """The reference under gloo DDP, world 2: AllGather_multi + DDP averaging semantics."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    [p.join() for p in procs]
    g0, g1 = res[0][4], res[1][4]
    for k in g0:  # DDP leaves identical averaged grads on both ranks
        assert np.allclose(g0[k], g1[k], atol=1e-6), k
    save("ddp2_tiny", loss1=np.array([res[0][1], res[1][1]]), loss2=np.array([res[0][2], res[1][2]]),
         grad_norm=res[0][3], seed=16, batch_seed0=100, batch_seed1=101, B=2, T=2, caption_len=8,
         **{"g_" + k: v for k, v in g0.items()})


class _PinnedHFAdamW(torch.optim.Optimizer):
    """transformers==4.10.2 `AdamW` (v2/requirement.txt:148; the package is absent here and the installed 5.x dropped the
    class), restated from its published algorithm as a torch.optim.Optimizer so that the REFERENCE's loop statements
    (`optimizer.zero_grad()`, `optimizer.step()`) drive it: a parameter whose `.grad` is None is skipped, one whose `.grad`
    is a zero tensor is updated (moments decay, the weights move by m / (sqrt(v) + eps) and by the decoupled decay);
    `state['step']` is per parameter."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st.update(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                st["step"] += 1
                st["exp_avg"].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                size = g["lr"] * (1.0 - b2 ** st["step"]) ** 0.5 / (1.0 - b1 ** st["step"])
                p.addcdiv_(st["exp_avg"], st["exp_avg_sq"].sqrt().add_(g["eps"]), value=-size)
                if g["weight_decay"] > 0.0:
                    p.add_(p, alpha=-g["lr"] * g["weight_decay"])


def alternating_arch():
    """the GPU tests' reduced architecture with the reference's text depth, so that the entrypoint's hard-coded
    `resblocks.9 .. 11` tuning rule (train_dist_TVTSv2_ViT_B_16.py:69) applies as written"""
    return O.tiny_arch(name="small", image=64, patch=16, width=256, heads=4, layers=2, embed=128, text_width=128,
                       text_heads=2, text_layers=12, text_tune_from=9, vocab=512, context=16)


def gen_alternating():
    """trainer.py:451-512 as the kept configs run it: loader 0 = YT-Temporal (NT = 4, sorting loss), loader 1 = WebVid
    (NT = 1, `pred_order is None`, loss2 = 0), ONE optimizer step per loader and loop iteration, `optimizer.zero_grad()` with
    no argument in front of each.  Under the pinned torch 1.11 (requirement.txt:142) zero_grad() keeps ZERO TENSORS
    (set_to_none=False was the default until 2.0), so in a WebVid step every `pred_model.*` tensor has g = 0, not None, and
    HF AdamW updates it.  Real reference modules, forward wiring, losses and autograd; parameter groups by executing the
    entrypoint's own grouping statements on the module; 3 loop iterations = 6 optimizer steps."""
    ns = import_reference()
    arch = alternating_arch()
    P = O.synth_params(arch, seed=21)
    m = TinyRefModel(ns, arch, P)
    src = open(os.path.join(REF, "train_dist_TVTSv2_ViT_B_16.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if "no_decay_names = [" in l)
    stop = next(i for i, l in enumerate(src) if "optimizer = transformers.AdamW" in l)
    body = "\n".join(l[4:] if l.startswith("    ") else l for l in src[start:stop])
    env = {"model": m}
    exec(compile(body, "<reference grouping>", "exec"), env)
    opt = _PinnedHFAdamW(env["optimizer_grouped_parameters"])
    assert [g["lr"] for g in opt.param_groups] == [1e-4, 1e-4, 1e-7, 1e-7]
    yt = [O.synth_batch(arch, B=4, T=2, seed=40 + i, caption_len=9) for i in range(3)]
    wv = [O.synth_batch(arch, B=4, T=3, seed=50 + i, n_trans=1, caption_len=9) for i in range(3)]
    ce = torch.nn.CrossEntropyLoss()
    l1s, l2s, none_after_first = [], [], None
    for it in range(3):
        for dl_idx, data in enumerate((yt[it], wv[it])):
            opt.zero_grad(set_to_none=False)  # == torch 1.11's optimizer.zero_grad()
            te, ve, pred = m(data)
            loss1 = ns.loss.NormSoftmaxLoss()(ns.m32.sim_matrix(ve, te))
            if pred is not None:
                loss2 = ce(pred.reshape(-1, pred.shape[-1]), data["label"].reshape(-1)) * 2
            else:
                loss2 = torch.Tensor([0])
            (loss1 + loss2).backward()
            opt.step()
            l1s.append(float(loss1.detach())); l2s.append(float(loss2.detach()))
            if none_after_first is None:
                none_after_first = [k for k, p in m.named_parameters() if p.requires_grad and p.grad is None]
    assert none_after_first == [], none_after_first  # the first step is a YT step: no trainable tensor is ever skipped
    pd = dict(m.named_parameters())
    assert all(opt.state[p]["step"] == 6 for g in opt.param_groups for p in g["params"])
    keep = ["pred_model.head.weight", "pred_model.head.bias", "pred_model.blocks.1.mlp.fc2.weight", "pred_model.type_embed",
            "video_model.transformer.resblocks.0.timeattn.proj.weight", "video_model.transformer.resblocks.1.ln_3.weight",
            "text_projection"]
    out = {}
    for k in keep:  # (matrices beyond the head: their leading 16 x 32 corner)
        cut = (lambda t: t[:16, :32]) if pd[k].dim() == 2 and pd[k].numel() > 4096 else (lambda t: t)
        out["p_" + k] = cut(pd[k])
        out["m_" + k] = cut(opt.state[pd[k]]["exp_avg"])
        out["v_" + k] = cut(opt.state[pd[k]]["exp_avg_sq"])
    save("alternating_steps", loss1=np.array(l1s), loss2=np.array(l2s), seed=21, yt_seeds=np.array([40, 41, 42]),
         wv_seeds=np.array([50, 51, 52]), B=4, T_yt=2, T_wv=3, caption_len=9, names=np.array(keep), **out)


GENS = {"block": gen_block, "vit": gen_vit, "text": gen_text, "sort": gen_sort, "model_tiny": gen_model_tiny,
        "model_b32": gen_model_b32, "groups": gen_groups, "ddp2": gen_ddp2, "model_h_tiny": gen_model_h_tiny, "model_h14": gen_model_h14, "metrics": gen_metrics, "downstream": gen_downstream, "transform": gen_transform, "transform_resize": gen_transform_resize, "tokenize": gen_tokenize, "downstream_more": gen_downstream_more, "model_b16": gen_model_b16, "ctor_init": gen_ctor_init, "alternating": gen_alternating}

if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(GENS)
    for w in which:
        with torch.enable_grad():
            GENS[w]()
