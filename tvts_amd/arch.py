"""Architecture tables and the parameter inventory (reference state-dict names).

Constructor arguments follow v2/model/model_dist_TVTSv2_ViT_{B_32,B_16,H_14}.py; the key order is the
reference's ``state_dict()`` order, which the checkpoint's optimizer state indexes into.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

ARCHS = {
    "B_32": dict(name="B_32", image=224, patch=32, width=768, heads=12, layers=12, embed=512,
                 text_width=512, text_heads=8, text_layers=12, text_tune_from=9, vocab=49408, context=77,
                 act="quick_gelu", tail="all_tokens", num_frames=12, mask_ratio=0.0,
                 sort_heads=8, sort_depth=2, n_trans=4),
    "B_16": dict(name="B_16", image=224, patch=16, width=768, heads=12, layers=12, embed=512,
                 text_width=512, text_heads=8, text_layers=12, text_tune_from=9, vocab=49408, context=77,
                 act="quick_gelu", tail="all_tokens", num_frames=12, mask_ratio=0.5,
                 sort_heads=8, sort_depth=2, n_trans=4),
    "H_14": dict(name="H_14", image=224, patch=14, width=1280, heads=16, layers=32, embed=1024,
                 text_width=1024, text_heads=16, text_layers=24, text_tune_from=18, vocab=49408, context=77,
                 act="gelu", tail="pooled_and_patches", num_frames=12, mask_ratio=0.7, block_order="openclip",
                 sort_heads=16, sort_depth=2, n_trans=4),
}


def small_arch(**over) -> dict:
    """A reduced architecture whose every dimension satisfies the HIP kernels' tiling constraints
    (head dim 64, GEMM K % 64 == 0, patch % 8 == 0); used by the GPU parity tests and smoke()."""
    a = dict(name="small", image=64, patch=16, width=256, heads=4, layers=2, embed=128,
             text_width=128, text_heads=2, text_layers=3, text_tune_from=1, vocab=512, context=16,
             act="quick_gelu", tail="all_tokens", num_frames=12, mask_ratio=0.5,
             sort_heads=2, sort_depth=2, n_trans=4)
    a.update(over)
    return a


def small_arch_h(**over) -> dict:
    """Reduced H/14-style architecture: head dim 80, 14x14 patches (K = 588, padded to 640 for the MFMA GEMM),
    erf-GELU, OpenCLIP block registration order, ln_post on CLS only, tube mask 0.7."""
    a = small_arch(name="H_14", image=56, patch=14, width=320, heads=4, act="gelu", tail="pooled_and_patches",
                   block_order="openclip", mask_ratio=0.7)
    a.update(over)
    return a


# ---- v1 TVTS: v1/model/model_dist_TVTS.py:38-47,62-76, v1/configs/dist-yt-pt.json,
#      v1/data_loader/YTTemporal_dataset.py:68 (mask ratio 0.75), distilbert-base-uncased text tower
ARCH_V1 = dict(name="v1", family="v1", image=224, patch=16, tubelet=2, width=768, heads=12, layers=12, num_frames=16,
               mask_ratio=0.75, text_width=768, text_heads=12, text_layers=6, text_ffn=3072, vocab=30522, max_pos=512,
               embed=256, sort_width=768, sort_heads=12, sort_depth=2, n_trans=4, act="gelu", tail="all_tokens")
ARCHS["v1"] = ARCH_V1


def small_arch_v1(**over) -> dict:
    """reduced v1 architecture inside the HIP kernels' tiling constraints (head dim 64, GEMM K % 64 == 0)"""
    a = dict(ARCH_V1, name="v1_small", image=64, width=128, heads=2, layers=2, num_frames=8, mask_ratio=0.5, text_width=128,
             text_heads=2, text_layers=2, text_ffn=256, vocab=1000, max_pos=40, embed=64, sort_width=128, sort_heads=2)
    a.update(over)
    return a


def param_shapes_v1(a) -> "OrderedDict[str, Tuple[int, ...]]":
    """state-dict keys of v1 `TVTS` in registration order: text_model (Hugging Face DistilBertModel), video_model, txt_proj,
    vid_proj, pred_model (model_dist_TVTS.py:34,58,62-76)"""
    o: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    Wt, Ff, W, E, Ws = a["text_width"], a["text_ffn"], a["width"], a["embed"], a["sort_width"]
    o["text_model.embeddings.word_embeddings.weight"] = (a["vocab"], Wt)
    o["text_model.embeddings.position_embeddings.weight"] = (a["max_pos"], Wt)
    o["text_model.embeddings.LayerNorm.weight"] = (Wt,)
    o["text_model.embeddings.LayerNorm.bias"] = (Wt,)
    for i in range(a["text_layers"]):
        p = f"text_model.transformer.layer.{i}."
        for lin in ("q_lin", "k_lin", "v_lin", "out_lin"):
            o[p + f"attention.{lin}.weight"] = (Wt, Wt)
            o[p + f"attention.{lin}.bias"] = (Wt,)
        o[p + "sa_layer_norm.weight"] = (Wt,)
        o[p + "sa_layer_norm.bias"] = (Wt,)
        o[p + "ffn.lin1.weight"] = (Ff, Wt)
        o[p + "ffn.lin1.bias"] = (Ff,)
        o[p + "ffn.lin2.weight"] = (Wt, Ff)
        o[p + "ffn.lin2.bias"] = (Wt,)
        o[p + "output_layer_norm.weight"] = (Wt,)
        o[p + "output_layer_norm.bias"] = (Wt,)
    o["video_model.cls_token"] = (1, 1, W)
    o["video_model.pos_embed"] = (1, patches_per_frame(a) + 1, W)
    o["video_model.temporal_embed"] = (1, a["num_frames"] // a["tubelet"], W)
    o["video_model.patch_embed.proj.weight"] = (W, 3, a["tubelet"], a["patch"], a["patch"])
    o["video_model.patch_embed.proj.bias"] = (W,)
    for i in range(a["layers"]):
        p = f"video_model.blocks.{i}."
        for k, shp in (("norm1.weight", (W,)), ("norm1.bias", (W,)), ("attn.qkv.weight", (3 * W, W)), ("attn.qkv.bias", (3 * W,)),
                       ("attn.proj.weight", (W, W)), ("attn.proj.bias", (W,)), ("norm2.weight", (W,)), ("norm2.bias", (W,)),
                       ("mlp.fc1.weight", (4 * W, W)), ("mlp.fc1.bias", (4 * W,)), ("mlp.fc2.weight", (W, 4 * W)),
                       ("mlp.fc2.bias", (W,))):
            o[p + k] = shp
    o["video_model.norm.weight"] = (W,)
    o["video_model.norm.bias"] = (W,)
    o["txt_proj.1.weight"] = (E, Wt)
    o["txt_proj.1.bias"] = (E,)
    o["vid_proj.0.weight"] = (E, W)
    o["vid_proj.0.bias"] = (E,)
    o["pred_model.type_embed"] = (1, 2, Ws)
    for i in range(a["sort_depth"]):
        p = f"pred_model.blocks.{i}."
        for k, shp in (("norm1.weight", (Ws,)), ("norm1.bias", (Ws,)), ("attn.qkv.weight", (3 * Ws, Ws)), ("attn.qkv.bias", (3 * Ws,)),
                       ("attn.proj.weight", (Ws, Ws)), ("attn.proj.bias", (Ws,)), ("norm2.weight", (Ws,)), ("norm2.bias", (Ws,)),
                       ("mlp.fc1.weight", (4 * Ws, Ws)), ("mlp.fc1.bias", (4 * Ws,)), ("mlp.fc2.weight", (Ws, 4 * Ws)),
                       ("mlp.fc2.bias", (Ws,))):
            o[p + k] = shp
    o["pred_model.norm.weight"] = (Ws,)
    o["pred_model.norm.bias"] = (Ws,)
    o["pred_model.head.weight"] = (a["n_trans"], Ws)
    o["pred_model.head.bias"] = (a["n_trans"],)
    return o


def patches_per_frame(arch) -> int:
    return (arch["image"] // arch["patch"]) ** 2


def n_keep(arch) -> int:
    # same float expression as v2/model/video_encoder_ViT_B_16.py:220
    return int(patches_per_frame(arch) * (1 - arch["mask_ratio"]))


def param_shapes(arch) -> "OrderedDict[str, Tuple[int, ...]]":
    if arch.get("family") == "v1":
        return param_shapes_v1(arch)
    W, E, Wt, p = arch["width"], arch["embed"], arch["text_width"], arch["patch"]
    o: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    o["text_positional_embedding"] = (arch["context"], Wt)
    o["text_projection"] = (Wt, E)
    openclip = arch.get("block_order") == "openclip"  # H/14: OpenCLIP registration order (ln_1 first)
    for i in range(arch["text_layers"]):
        pre = f"text_model.resblocks.{i}."
        blk = {"attn.in_proj_weight": (3 * Wt, Wt), "attn.in_proj_bias": (3 * Wt,), "attn.out_proj.weight": (Wt, Wt),
               "attn.out_proj.bias": (Wt,), "ln_1.weight": (Wt,), "ln_1.bias": (Wt,), "mlp.c_fc.weight": (4 * Wt, Wt),
               "mlp.c_fc.bias": (4 * Wt,), "mlp.c_proj.weight": (Wt, 4 * Wt), "mlp.c_proj.bias": (Wt,),
               "ln_2.weight": (Wt,), "ln_2.bias": (Wt,)}
        order = list(blk)
        if openclip:  # v2/OpenCLIP/transformer.py:189-216: ln_1, attn, ln_2, mlp
            order = ["ln_1.weight", "ln_1.bias"] + [k for k in order if k.startswith("attn.")] + \
                    ["ln_2.weight", "ln_2.bias"] + [k for k in order if k.startswith("mlp.")]
        for k in order:
            o[pre + k] = blk[k]
    o["text_token_embedding.weight"] = (arch["vocab"], Wt)
    o["text_ln_final.weight"] = (Wt,)
    o["text_ln_final.bias"] = (Wt,)
    o["video_model.class_embedding"] = (W,)
    o["video_model.positional_embedding"] = (patches_per_frame(arch) + 1, W)
    o["video_model.proj"] = (W, E)
    o["video_model.temporal_embedding"] = (arch["num_frames"], W)
    o["video_model.conv1.weight"] = (W, 3, p, p)
    o["video_model.ln_pre.weight"] = (W,)
    o["video_model.ln_pre.bias"] = (W,)
    for i in range(arch["layers"]):
        pre = f"video_model.transformer.resblocks.{i}."
        blk = {}
        for a in ("attn", "timeattn"):
            blk[a + ".qkv.weight"] = (3 * W, W)
            blk[a + ".qkv.bias"] = (3 * W,)
            blk[a + ".proj.weight"] = (W, W)
            blk[a + ".proj.bias"] = (W,)
        for ln in ("ln_3", "ln_1"):
            blk[ln + ".weight"] = (W,)
            blk[ln + ".bias"] = (W,)
        blk["mlp.c_fc.weight"] = (4 * W, W)
        blk["mlp.c_fc.bias"] = (4 * W,)
        blk["mlp.c_proj.weight"] = (W, 4 * W)
        blk["mlp.c_proj.bias"] = (W,)
        blk["ln_2.weight"] = (W,)
        blk["ln_2.bias"] = (W,)
        order = list(blk)  # B models (video_encoder_ViT_B_16.py:98-111): attn, timeattn, ln_3, ln_1, mlp, ln_2
        if openclip:       # H/14 (video_encoder_ViT_H_14.py:221-240): ln_1, attn, timeattn, ln_3, ln_2, mlp
            pick = lambda pfx: [k for k in blk if k.startswith(pfx)]  # noqa: E731
            order = pick("ln_1.") + pick("attn.") + pick("timeattn.") + pick("ln_3.") + pick("ln_2.") + pick("mlp.")
        for k in order:
            o[pre + k] = blk[k]
    o["video_model.ln_post.weight"] = (W,)
    o["video_model.ln_post.bias"] = (W,)
    if arch.get("sort_head", True):  # the downstream inference models carry no transcript-sorting head
        o["pred_model.type_embed"] = (1, 2, E)
        for i in range(arch["sort_depth"]):
            pre = f"pred_model.blocks.{i}."
            o[pre + "norm1.weight"] = (E,)
            o[pre + "norm1.bias"] = (E,)
            o[pre + "attn.qkv.weight"] = (3 * E, E)
            o[pre + "attn.qkv.bias"] = (3 * E,)
            o[pre + "attn.proj.weight"] = (E, E)
            o[pre + "attn.proj.bias"] = (E,)
            o[pre + "norm2.weight"] = (E,)
            o[pre + "norm2.bias"] = (E,)
            o[pre + "mlp.fc1.weight"] = (4 * E, E)
            o[pre + "mlp.fc1.bias"] = (4 * E,)
            o[pre + "mlp.fc2.weight"] = (E, 4 * E)
            o[pre + "mlp.fc2.bias"] = (E,)
        o["pred_model.norm.weight"] = (E,)
        o["pred_model.norm.bias"] = (E,)
        o["pred_model.head.weight"] = (arch["n_trans"], E)
        o["pred_model.head.bias"] = (arch["n_trans"],)
    return o


def is_mfma_weight(name: str, shape) -> bool:
    """Parameters consumed by the bf16 MFMA GEMMs (they get bf16 shadows, plain and transposed)."""
    if name in ("video_model.proj", "video_model.conv1.weight", "video_model.patch_embed.proj.weight"):
        return True
    if name in ("txt_proj.1.weight", "vid_proj.0.weight"):  # v1 projections of [N, W] rows: the small fp32 kernel
        return False
    if name.startswith("pred_model.head"):
        return False
    return len(shape) == 2 and name.endswith(("weight", "in_proj_weight")) and "embedding" not in name


# optimizer grouping: v2/train_dist_TVTSv2_ViT_B_16.py:66-107 (H/14: train_dist_TVTSv2_ViT_H_14.py:68-71)
GROUP_HPARAMS = ((1e-4, 0.05), (1e-4, 0.0), (1e-7, 0.05), (1e-7, 0.0))


def param_group_of(name: str, arch) -> int:
    """0 new-decay, 1 new-nodecay, 2 clip-decay, 3 clip-nodecay, -1 frozen (requires_grad False).
    v1: ONE group, every parameter (v1/train_dist_TVTS.py:68-69, configs/dist-yt-pt.json: AdamW lr 1e-4, weight_decay 0)."""
    if arch.get("family") == "v1":
        return 0
    no_decay = ["bias", "LayerNorm", "ln_", "norm"]
    if arch["name"] == "H_14":
        no_decay += ["ls_", "LayerScale"]
    nd = any(s in name for s in no_decay)
    if "video_model" in name:
        new = "timeattn" in name or "ln_3" in name
        return (0 if new else 2) + int(nd)
    if "text" in name:
        if "resblocks" in name:
            tune = ["resblocks.%d." % i for i in range(arch["text_tune_from"], arch["text_layers"])]
            if not any(t in name for t in tune):
                return -1
        return 2 + int(nd)
    return int(nd)
