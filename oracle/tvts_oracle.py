"""CPU oracle for the TVTSv2 pretrain step -- TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 torch-CPU restatement of the reference algorithm for
the hot path named in BASELINE.json.  It is *not* part of
the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker.  ``tvts_amd`` never imports it and fails loudly when the HIP library is
missing.

Parity status: PINNED against outputs of the reference itself, imported in the
build container by ``tests/golden/make_golden.py`` (the reference has no golden
vectors of its own).  The fixtures live in
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against
every one of them.  Exception: the optimizer.  ``transformers.AdamW`` (pinned
``transformers==4.10.2`` in ``v2/requirement.txt``) is a third-party dependency
that is absent from the reference tree and from this image, so
``hf_adamw_step`` restates the published algorithm and is "parity unpinned".

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  The formulation is index-based (explicit token index
tables) rather than the reference's einops regrouping, so the two share no code.

Parameters are passed as a flat ``dict[str, Tensor]`` that uses the reference
state-dict key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# architecture table (v2/model/model_dist_TVTSv2_ViT_{B_32,B_16,H_14}.py ctor arguments)
# --------------------------------------------------------------------------------------

ARCHS = {
    # v2/model/model_dist_TVTSv2_ViT_B_32.py:29-31, CLIP ViT-B/32 text tower 512/8/12
    "B_32": dict(name="B_32", image=224, patch=32, width=768, heads=12, layers=12, embed=512,
                 text_width=512, text_heads=8, text_layers=12, text_tune_from=9, vocab=49408, context=77,
                 act="quick_gelu", tail="all_tokens", num_frames=12, mask_ratio=0.0,
                 sort_heads=8, sort_depth=2, n_trans=4),
    # v2/model/model_dist_TVTSv2_ViT_B_16.py:29-31
    "B_16": dict(name="B_16", image=224, patch=16, width=768, heads=12, layers=12, embed=512,
                 text_width=512, text_heads=8, text_layers=12, text_tune_from=9, vocab=49408, context=77,
                 act="quick_gelu", tail="all_tokens", num_frames=12, mask_ratio=0.5,
                 sort_heads=8, sort_depth=2, n_trans=4),
    # v2/model/model_dist_TVTSv2_ViT_H_14.py:44-83, OpenCLIP/model_configs/ViT-H-14.json
    "H_14": dict(name="H_14", image=224, patch=14, width=1280, heads=16, layers=32, embed=1024,
                 text_width=1024, text_heads=16, text_layers=24, text_tune_from=18, vocab=49408, context=77,
                 act="gelu", tail="pooled_and_patches", num_frames=12, mask_ratio=0.7, block_order="openclip",
                 sort_heads=16, sort_depth=2, n_trans=4),
}


def tiny_arch(**over) -> dict:
    """A miniature architecture with the same structure, for fast unit parity."""
    a = dict(name="tiny", image=32, patch=8, width=64, heads=2, layers=2, embed=32,
             text_width=32, text_heads=2, text_layers=3, text_tune_from=1, vocab=97, context=12,
             act="quick_gelu", tail="all_tokens", num_frames=12, mask_ratio=0.5,
             sort_heads=2, sort_depth=2, n_trans=4)
    a.update(over)
    return a


def patches_per_frame(arch) -> int:
    return (arch["image"] // arch["patch"]) ** 2


def n_keep(arch) -> int:
    # v2/model/video_encoder_ViT_B_16.py:220 -- same float expression as the reference
    return int(patches_per_frame(arch) * (1 - arch["mask_ratio"]))


# --------------------------------------------------------------------------------------
# parameter inventory + deterministic synthetic parameters / batches
# --------------------------------------------------------------------------------------

def param_shapes(arch) -> "Dict[str, Tuple[int, ...]]":
    """State-dict key -> shape, in the reference's registration order."""
    W, E, Wt = arch["width"], arch["embed"], arch["text_width"]
    p = arch["patch"]
    out: Dict[str, Tuple[int, ...]] = {}
    out["text_positional_embedding"] = (arch["context"], Wt)
    out["text_projection"] = (Wt, E)
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
            out[pre + k] = blk[k]
    out["text_token_embedding.weight"] = (arch["vocab"], Wt)
    out["text_ln_final.weight"] = (Wt,)
    out["text_ln_final.bias"] = (Wt,)
    out["video_model.class_embedding"] = (W,)
    out["video_model.positional_embedding"] = (patches_per_frame(arch) + 1, W)
    out["video_model.proj"] = (W, E)
    out["video_model.temporal_embedding"] = (arch["num_frames"], W)
    out["video_model.conv1.weight"] = (W, 3, p, p)
    out["video_model.ln_pre.weight"] = (W,)
    out["video_model.ln_pre.bias"] = (W,)
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
            out[pre + k] = blk[k]
    out["video_model.ln_post.weight"] = (W,)
    out["video_model.ln_post.bias"] = (W,)
    if arch.get("sort_head", True):  # the downstream inference models carry no transcript-sorting head
        out["pred_model.type_embed"] = (1, 2, E)
        for i in range(arch["sort_depth"]):
            pre = f"pred_model.blocks.{i}."
            out[pre + "norm1.weight"] = (E,)
            out[pre + "norm1.bias"] = (E,)
            out[pre + "attn.qkv.weight"] = (3 * E, E)
            out[pre + "attn.qkv.bias"] = (3 * E,)
            out[pre + "attn.proj.weight"] = (E, E)
            out[pre + "attn.proj.bias"] = (E,)
            out[pre + "norm2.weight"] = (E,)
            out[pre + "norm2.bias"] = (E,)
            out[pre + "mlp.fc1.weight"] = (4 * E, E)
            out[pre + "mlp.fc1.bias"] = (4 * E,)
            out[pre + "mlp.fc2.weight"] = (E, 4 * E)
            out[pre + "mlp.fc2.bias"] = (E,)
        out["pred_model.norm.weight"] = (E,)
        out["pred_model.norm.bias"] = (E,)
        out["pred_model.head.weight"] = (arch["n_trans"], E)
        out["pred_model.head.bias"] = (arch["n_trans"],)
    return out


def _key_seed(seed: int, name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def synth_params(arch, seed: int = 0) -> Params:
    """Deterministic per-key synthetic parameters (independent of module construction order).

    Scales mimic a trained CLIP checkpoint's order of magnitude; ``timeattn`` is drawn
    N(0, 0.02^2) because at reference init it is numerically dead.
    LayerNorm gains are 1 + small noise, biases small noise, so every term is exercised.
    """
    out: Params = {}
    for name, shape in param_shapes(arch).items():
        g = torch.Generator().manual_seed(_key_seed(seed, name))
        t = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = any(s in name for s in ("ln_", "norm"))
        if is_norm and leaf == "weight":
            t = 1.0 + 0.05 * t
        elif leaf in ("bias", "in_proj_bias"):
            t = 0.02 * t
        elif name == "text_token_embedding.weight":
            t = 0.02 * t
        elif name == "text_positional_embedding":
            t = 0.01 * t
        elif name in ("video_model.class_embedding", "video_model.positional_embedding",
                      "video_model.temporal_embedding", "video_model.proj"):
            t = t * arch["width"] ** -0.5
        elif name == "text_projection":
            t = t * arch["text_width"] ** -0.5
        elif name == "pred_model.type_embed":
            t = 0.02 * t
        elif name == "video_model.conv1.weight":
            t = t * (3 * arch["patch"] ** 2) ** -0.5
        elif "timeattn" in name:
            t = 0.02 * t
        else:  # dense weights [out, in]
            t = t * (shape[-1] ** -0.5) * 0.7
        out[name] = t
    return out


def synth_clip_state_dict(layout, seed: int = 0) -> Dict[str, Tensor]:
    """A seeded stand-in for a pretrained CLIP / OpenCLIP state dict: one N(0, 0.02^2) tensor per (key, shape) of ``layout``,
    drawn per key (independent of key order) like synth_params.  The constructor-initialisation fixtures
    (tests/golden/ctor_init_*.npz, make_golden.py::gen_ctor_init) feed it to the REAL reference constructors; the tests feed the
    same tensors to this package's constructor (v2/model/model_dist_TVTSv2_ViT_B_16.py:19-45, ..._H_14.py:44-83)."""
    out: Dict[str, Tensor] = {}
    for name, shape in layout.items():
        g = torch.Generator().manual_seed(_key_seed(seed, "clip/" + name))
        out[name] = 0.02 * torch.randn(tuple(int(x) for x in shape), generator=g, dtype=torch.float32)
    return out


def tensor_crc(t: Tensor) -> int:
    """CRC-32 of a tensor's fp32 bytes: bit-identity check for the fixtures that cannot carry 10^8-element tensors"""
    import zlib
    return zlib.crc32(t.detach().to(torch.float32).contiguous().cpu().numpy().tobytes())


def synth_batch(arch, B: int, T: int, seed: int = 0, n_trans: Optional[int] = None,
                caption_len: int = 32) -> Dict[str, Tensor]:
    """Synthetic clip-caption batch in the reference's batch-dict contract."""
    NT = arch["n_trans"] if n_trans is None else n_trans
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, T, 3, arch["image"], arch["image"], generator=g, dtype=torch.float32)
    L = arch["context"]
    cl = min(caption_len, L)
    text = torch.zeros(NT * B, L, dtype=torch.int32)
    sot, eot = arch["vocab"] - 2, arch["vocab"] - 1
    text[:, 0] = sot
    text[:, 1:cl - 1] = torch.randint(1, arch["vocab"] - 408 if arch["vocab"] > 1000 else arch["vocab"] - 2,
                                      (NT * B, cl - 2), generator=g, dtype=torch.int32)
    text[:, cl - 1] = eot
    ppf, n = patches_per_frame(arch), n_keep(arch)
    keep = torch.stack([torch.randperm(ppf, generator=g)[:n] for _ in range(B)]).to(torch.int64)
    batch = {"video": video, "text": text, "keep_ind": keep}
    if NT == arch["n_trans"]:
        batch["label"] = torch.arange(arch["n_trans"]).repeat(B, 1)
    return batch


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------

def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    # v2/model/video_encoder_ViT_B_16.py:79-85 (fp32 LayerNorm), sort_transformer.py:99 (eps 1e-6)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def quick_gelu(x: Tensor) -> Tensor:
    # v2/model/video_encoder_ViT_B_16.py:88-90
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x: Tensor) -> Tensor:
    # nn.GELU default (sort_transformer.py:17; H/14 act_layer)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def _act(arch):
    return quick_gelu if arch["act"] == "quick_gelu" else gelu_erf


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def fake_quant_e4m3(t: Tensor, rowwise: bool = False) -> Tensor:
    """OCP e4m3 quantise -> dequantise (scale = amax / 448, round to nearest even, gradient passed straight through): what
    the fp8 weight / activation GEMMs of BASELINE config 4 see.  One scale for the tensor (weights), or with `rowwise` one per
    row of the last dimension (activations: a scale per token)."""
    d = t.detach()
    if rowwise:
        amax = d.abs().amax(dim=-1, keepdim=True)
        scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    else:
        amax = d.abs().max()
        scale = amax / 448.0 if float(amax) > 0 else torch.ones(())
    q = (d / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    return t + (q - t).detach()


def linear_fp8(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Linear whose FORWARD product runs on e4m3 copies of the activation and the weight (the backward of the build
    keeps bf16 operands, i.e. straight-through here)."""
    y = fake_quant_e4m3(x, rowwise=True) @ fake_quant_e4m3(w).t()
    return y if b is None else y + b


def _softmax_attend(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T) v over the last two dims (v2/model/video_encoder_ViT_B_16.py:11-15)."""
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    return torch.softmax(s, dim=-1) @ v


# --------------------------------------------------------------------------------------
# divided space-time attention (v2/model/video_encoder_ViT_B_16.py:38-76)
# --------------------------------------------------------------------------------------

def divided_attention(x: Tensor, wqkv: Tensor, bqkv: Tensor, wo: Tensor, bo: Tensor,
                      heads: int, mode: str, T: int, n: int, lin=None) -> Tensor:
    """VarAttention.forward restated with explicit token tables.

    x: [B, 1+T*n, W].  mode 'time': token (f, i) attends {CLS} + {(f', i)}; mode 'space':
    token (f, i) attends {CLS} + {(f, i')}.  The CLS query attends all tokens.  q is scaled by
    dh^-0.5 before the CLS split (:45-51); CLS k/v are key 0 of every group (:56-60).
    """
    lin = linear if lin is None else lin
    out = divided_attention_core(lin(x, wqkv, bqkv), heads, mode, T, n)
    return lin(out, wo, bo)


def divided_attention_core(qkv_packed: Tensor, heads: int, mode: str, T: int, n: int) -> Tensor:
    """The attention proper on a packed [B, S, 3W] (q | k | v) tensor -> merged heads [B, S, W]."""
    B, S, W3 = qkv_packed.shape
    W = W3 // 3
    dh = W // heads
    qkv = qkv_packed.reshape(B, S, 3, heads, dh)
    q = qkv[:, :, 0].permute(0, 2, 1, 3) * dh ** -0.5  # [B,h,S,dh]
    k = qkv[:, :, 1].permute(0, 2, 1, 3)
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    cls_out = _softmax_attend(q[:, :, 0:1], k, v)  # [B,h,1,dh]

    def grid(t):  # patch tokens -> [B,h,T,n,dh]
        return t[:, :, 1:].reshape(B, heads, T, n, dh)

    qg, kg, vg = grid(q), grid(k), grid(v)
    if mode == "time":  # groups over i, sequence over f
        qg, kg, vg = (t.transpose(2, 3) for t in (qg, kg, vg))  # [B,h,n,T,dh]
    G = qg.shape[2]
    kc = k[:, :, 0:1].unsqueeze(2).expand(B, heads, G, 1, dh)
    vc = v[:, :, 0:1].unsqueeze(2).expand(B, heads, G, 1, dh)
    out = _softmax_attend(qg, torch.cat([kc, kg], 3), torch.cat([vc, vg], 3))
    if mode == "time":
        out = out.transpose(2, 3)
    out = torch.cat([cls_out, out.reshape(B, heads, T * n, dh)], 2)  # [B,h,S,dh]
    return out.permute(0, 2, 1, 3).reshape(B, S, W)


def st_block(x: Tensor, P: Params, pre: str, arch, T: int, n: int) -> Tensor:
    """ResidualSpaceTimeAttentionBlock.forward (v2/model/video_encoder_ViT_B_16.py:113-124).

    NB the space residual starts from the block input x, not from the time residual (:121).
    """
    h = arch["heads"]
    lin = linear_fp8 if arch.get("fp8") else linear
    t_out = divided_attention(layer_norm(x, P[pre + "ln_3.weight"], P[pre + "ln_3.bias"], 1e-5),
                              P[pre + "timeattn.qkv.weight"], P[pre + "timeattn.qkv.bias"],
                              P[pre + "timeattn.proj.weight"], P[pre + "timeattn.proj.bias"], h, "time", T, n, lin)
    t_res = x + t_out
    s_out = divided_attention(layer_norm(t_res, P[pre + "ln_1.weight"], P[pre + "ln_1.bias"], 1e-5),
                              P[pre + "attn.qkv.weight"], P[pre + "attn.qkv.bias"],
                              P[pre + "attn.proj.weight"], P[pre + "attn.proj.bias"], h, "space", T, n, lin)
    s_res = x + s_out
    hid = _act(arch)(lin(layer_norm(s_res, P[pre + "ln_2.weight"], P[pre + "ln_2.bias"], 1e-5),
                         P[pre + "mlp.c_fc.weight"], P[pre + "mlp.c_fc.bias"]))
    return s_res + lin(hid, P[pre + "mlp.c_proj.weight"], P[pre + "mlp.c_proj.bias"])


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # v2/video_transforms/videoaug.py:17,26


def frames_to_video(frames_u8: Tensor, image: int, crop: Optional[Tensor] = None) -> Tensor:
    """uint8 frames [B, T, H0, W0, 3] (decoded + resized) -> the fp32 [B, T, 3, image, image] clip the model is fed:
    CenterCrop / RandomCrop offset, ClipToTensor (float32, / 255) and Normalize ((v - mean) / std in fp32), in the
    reference's operation order (video_transforms/video_transform.py:24-75, functional.py:81-97, base_dataset.py:125-127)."""
    B, T, H0, W0, _ = frames_u8.shape
    out = torch.empty(B, T, 3, image, image, dtype=torch.float32)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32)[:, None, None]
    for b in range(B):
        # CenterCrop: int(round((im - size) / 2.)) -- Python's round, i.e. half to even (video_transform.py:454-455)
        y0, x0 = (int(round((H0 - image) / 2.)), int(round((W0 - image) / 2.))) if crop is None else (int(crop[b, 0]), int(crop[b, 1]))
        clip = frames_u8[b, :, y0:y0 + image, x0:x0 + image, :].permute(0, 3, 1, 2).float().div(255)
        out[b] = clip.sub(mean).div(std)
    return out


def video_embed_tokens(P: Params, video: Tensor, keep_ind: Tensor, arch) -> Tensor:
    """Patch embed + CLS + space/time position + tube-mask gather + ln_pre
    (v2/model/video_encoder_ViT_B_16.py:176-218).  Gather-then-embed, which is
    output-equivalent to the reference's embed-then-gather."""
    B, T = video.shape[:2]
    p, W = arch["patch"], arch["width"]
    g = arch["image"] // p
    n = keep_ind.shape[1]
    # patch pixel vectors in conv-weight order (c, py, px): [B,T,ppf,3*p*p]
    pix = video.reshape(B, T, 3, g, p, g, p).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, g * g, 3 * p * p)
    idx = keep_ind.to(torch.int64)[:, None, :, None].expand(B, T, n, 3 * p * p)
    kept = torch.gather(pix, 2, idx)  # [B,T,n,3pp]
    tok = kept @ P["video_model.conv1.weight"].reshape(W, -1).t()  # [B,T,n,W]
    pos = P["video_model.positional_embedding"]
    tok = tok + pos[1:][keep_ind.to(torch.int64)][:, None] + P["video_model.temporal_embedding"][:T][None, :, None]
    cls = (P["video_model.class_embedding"] + pos[0]).expand(B, 1, W)
    x = torch.cat([cls, tok.reshape(B, T * n, W)], 1)
    return layer_norm(x, P["video_model.ln_pre.weight"], P["video_model.ln_pre.bias"], 1e-5)


def video_tower(P: Params, video: Tensor, keep_ind: Tensor, arch,
                taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """VisionTransformer.forward -> (video_embedding [B,E], sort-head tokens [B,S',E]).

    tail 'all_tokens' (B models, video_encoder_ViT_B_16.py:229-235 + model_dist..B_16.py:113-116):
    ln_post on all S tokens, @proj, CLS row is the embedding, all S rows go to the sort head.
    tail 'pooled_and_patches' (H/14, video_encoder_ViT_H_14.py:472-484): ln_post on CLS only,
    patch tokens projected without LN and without CLS.
    """
    if video.dim() == 4:
        video = video.unsqueeze(1)
    T = video.shape[1]
    n = keep_ind.shape[1]
    if keep_ind.shape[0] == 1 and video.shape[0] > 1:  # one tube mask broadcast over the batch (downstream scripts)
        keep_ind = keep_ind.expand(video.shape[0], -1)
    x = video_embed_tokens(P, video, keep_ind, arch)
    if taps is not None:
        taps["vit_in"] = x
    for i in range(arch["layers"]):
        x = st_block(x, P, f"video_model.transformer.resblocks.{i}.", arch, T, n)
        if taps is not None:
            taps[f"vit_block{i}"] = x
    lw, lb, proj = P["video_model.ln_post.weight"], P["video_model.ln_post.bias"], P["video_model.proj"]
    if arch["tail"] == "all_tokens":
        out = layer_norm(x, lw, lb, 1e-5) @ proj
        return out[:, 0], out
    pooled = layer_norm(x[:, 0], lw, lb, 1e-5) @ proj
    return pooled, x[:, 1:] @ proj


# --------------------------------------------------------------------------------------
# CLIP text tower (v2/CLIP/clip/model.py:171-203,330-358; model_dist..B_16.py:97-111)
# --------------------------------------------------------------------------------------

def text_block(x: Tensor, P: Params, pre: str, heads: int, act) -> Tensor:
    """ResidualAttentionBlock with causal nn.MultiheadAttention, x: [N, L, Wt]."""
    N, L, Wt = x.shape
    dh = Wt // heads
    y = layer_norm(x, P[pre + "ln_1.weight"], P[pre + "ln_1.bias"], 1e-5)
    qkv = linear(y, P[pre + "attn.in_proj_weight"], P[pre + "attn.in_proj_bias"]).reshape(N, L, 3, heads, dh)
    q = qkv[:, :, 0].permute(0, 2, 1, 3) * dh ** -0.5
    k = qkv[:, :, 1].permute(0, 2, 1, 3)
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    mask = torch.full((L, L), float("-inf")).triu(1)  # model.py:330-336
    o = _softmax_attend(q, k, v, mask).permute(0, 2, 1, 3).reshape(N, L, Wt)
    x = x + linear(o, P[pre + "attn.out_proj.weight"], P[pre + "attn.out_proj.bias"])
    y = layer_norm(x, P[pre + "ln_2.weight"], P[pre + "ln_2.bias"], 1e-5)
    y = act(linear(y, P[pre + "mlp.c_fc.weight"], P[pre + "mlp.c_fc.bias"]))
    return x + linear(y, P[pre + "mlp.c_proj.weight"], P[pre + "mlp.c_proj.bias"])


def text_tower(P: Params, ids: Tensor, arch, truncate: bool = True, taps: Optional[dict] = None) -> Tensor:
    """compute_text: ids int [N, context] -> [N, E] (row at argmax(ids) = EOT, @ text_projection).

    With truncate=True the context is cut to max(EOT)+1, exact under the causal mask
    (SURVEY.md App. B #14).
    """
    ids = ids.to(torch.int64)
    eot = ids.argmax(dim=-1)
    L = int(eot.max()) + 1 if truncate else ids.shape[1]
    x = P["text_token_embedding.weight"][ids[:, :L]] + P["text_positional_embedding"][:L]
    act = _act(arch)
    for i in range(arch["text_layers"]):
        x = text_block(x, P, f"text_model.resblocks.{i}.", arch["text_heads"], act)
        if taps is not None:
            taps[f"text_block{i}"] = x
    rows = x[torch.arange(x.shape[0]), eot]
    rows = layer_norm(rows, P["text_ln_final.weight"], P["text_ln_final.bias"], 1e-5)
    return rows @ P["text_projection"]


# --------------------------------------------------------------------------------------
# transcript sorting head (v2/model/sort_transformer.py:35-142)
# --------------------------------------------------------------------------------------

def sort_head(P: Params, text: Tensor, tokens: Tensor, arch) -> Tensor:
    """SortTransformer.forward(text [B,NT,E] (detached), tokens [B,S',E]) -> [B,NT,n_trans]."""
    E, h = arch["embed"], arch["sort_heads"]
    dh = E // h
    te = P["pred_model.type_embed"]
    x = torch.cat([tokens + te[:, 0], text + te[:, 1]], 1)
    B, So, _ = x.shape
    for i in range(arch["sort_depth"]):
        pre = f"pred_model.blocks.{i}."
        y = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], 1e-6)
        qkv = linear(y, P[pre + "attn.qkv.weight"], P[pre + "attn.qkv.bias"]).reshape(B, So, 3, h, dh)
        q = qkv[:, :, 0].permute(0, 2, 1, 3) * dh ** -0.5
        o = _softmax_attend(q, qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3))
        x = x + linear(o.permute(0, 2, 1, 3).reshape(B, So, E), P[pre + "attn.proj.weight"], P[pre + "attn.proj.bias"])
        y = layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], 1e-6)
        y = gelu_erf(linear(y, P[pre + "mlp.fc1.weight"], P[pre + "mlp.fc1.bias"]))
        x = x + linear(y, P[pre + "mlp.fc2.weight"], P[pre + "mlp.fc2.bias"])
    y = layer_norm(x[:, tokens.shape[1]:], P["pred_model.norm.weight"], P["pred_model.norm.bias"], 1e-6)
    return linear(y, P["pred_model.head.weight"], P["pred_model.head.bias"])


# --------------------------------------------------------------------------------------
# model forward + losses (model_dist..B_16.py:61-127, loss.py:13-25, trainer.py:479-496)
# --------------------------------------------------------------------------------------

def model_forward(P: Params, batch: dict, arch, truncate_text: bool = True,
                  taps: Optional[dict] = None) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """TVTSv2_*.forward(data) -> (text_emb [B,E], video_emb [B,E], pred_order [B,NT,4] | None).

    Captions are clip-major rows i*B+b (trainer.py:465-472); the sort head sees the *detached*
    per-caption embeddings (:69) and the contrastive text embedding is their mean (:74-76).
    """
    video = batch["video"]
    B = video.shape[0]
    t = text_tower(P, batch["text"], arch, truncate_text, taps)
    NT = t.shape[0] // B
    t = t.reshape(NT, B, -1)
    text_before = t.detach().permute(1, 0, 2)
    text_emb = t.mean(0)
    video_emb, tokens = video_tower(P, video, batch["keep_ind"], arch, taps)
    pred = sort_head(P, text_before, tokens, arch) if (NT != 1 and arch.get("sort_head", True)) else None
    return text_emb, video_emb, pred


def sim_matrix(a: Tensor, b: Tensor, eps: float = 1e-8) -> Tensor:
    # model_dist..B_16.py:119-127
    an = a / a.norm(dim=1, keepdim=True).clamp_min(eps)
    bn = b / b.norm(dim=1, keepdim=True).clamp_min(eps)
    return an @ bn.t()


def norm_softmax_loss(sim: Tensor, temperature: float = 0.05) -> Tensor:
    # loss.py:13-25 : both directions, no 1/2 factor
    x = sim / temperature
    li = torch.diagonal(x - torch.logsumexp(x, dim=1, keepdim=True)).mean()
    lj = torch.diagonal(x.t() - torch.logsumexp(x.t(), dim=1, keepdim=True)).mean()
    return -li - lj


def sorting_ce(pred: Tensor, label: Tensor) -> Tensor:
    # trainer.py:487-492 : 2 * CrossEntropy(mean)
    lg = pred.reshape(-1, pred.shape[-1])
    lb = label.reshape(-1).to(torch.int64)
    lse = torch.logsumexp(lg, dim=1)
    return 2.0 * (lse - lg[torch.arange(lg.shape[0]), lb]).mean()


def step_losses(P: Params, batch: dict, arch, truncate_text: bool = True):
    """One rank, world 1: (loss1, loss2, text_emb, video_emb, pred)."""
    te, ve, pred = model_forward(P, batch, arch, truncate_text)
    loss1 = norm_softmax_loss(sim_matrix(ve, te))
    loss2 = sorting_ce(pred, batch["label"]) if pred is not None else torch.zeros(())
    return loss1, loss2, te, ve, pred


def multi_rank_step(P: Params, batches: List[dict], arch, truncate_text: bool = True):
    """Single-process emulation of the W-rank step.

    Every rank computes the same global InfoNCE over the gathered embeddings;
    AllGather_multi.backward keeps the local rows only (trainer.py:52-57) and DDP averages,
    so the update gradient is grad( L1_global / W + mean_r L2_r ).  Returns
    (total_for_backward, loss1_global, [loss2_r]).
    """
    Wn = len(batches)
    tes, ves, l2 = [], [], []
    for b in batches:
        te, ve, pred = model_forward(P, b, arch, truncate_text)
        tes.append(te)
        ves.append(ve)
        l2.append(sorting_ce(pred, b["label"]) if pred is not None else torch.zeros(()))
    loss1 = norm_softmax_loss(sim_matrix(torch.cat(ves), torch.cat(tes)))
    total = loss1 / Wn + sum(l2) / Wn
    return total, loss1, l2


# --------------------------------------------------------------------------------------
# optimizer: parameter grouping + HF AdamW  (train_dist_TVTSv2_ViT_B_16.py:66-125)
# --------------------------------------------------------------------------------------

GROUP_HPARAMS = (  # (lr, weight_decay) for new-decay, new-nodecay, clip-decay, clip-nodecay (:118-123)
    (1e-4, 0.05), (1e-4, 0.0), (1e-7, 0.05), (1e-7, 0.0))


def param_groups(names: List[str], arch) -> Tuple[List[List[str]], List[str]]:
    """Name-substring grouping of train_dist_TVTSv2_ViT_B_16.py:66-107.

    Returns ([new_decay, new_nodecay, clip_decay, clip_nodecay], frozen)."""
    no_decay = ["bias", "LayerNorm", "ln_", "norm"]
    if arch["name"] == "H_14":  # train_dist_TVTSv2_ViT_H_14.py:68
        no_decay = no_decay + ["ls_", "LayerScale"]
    tune = ["resblocks.%d." % i for i in range(arch["text_tune_from"], arch["text_layers"])]
    groups: List[List[str]] = [[], [], [], []]
    frozen: List[str] = []
    for name in names:
        nd = any(s in name for s in no_decay)
        if "video_model" in name:
            new = "timeattn" in name or "ln_3" in name
            groups[(0 if new else 2) + (1 if nd else 0)].append(name)
        elif "text" in name:
            if "resblocks" in name and not any(t in name for t in tune):
                frozen.append(name)
            else:
                groups[2 + (1 if nd else 0)].append(name)
        else:
            groups[1 if nd else 0].append(name)
    return groups, frozen


def hf_adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
                  beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-6) -> None:
    """transformers.AdamW.step for one tensor, in place (transformers==4.10.2, optimization.py,
    class AdamW; absent from the reference tree -> restated from the published algorithm,
    PARITY UNPINNED).  eps is added outside the bias correction; the decoupled decay
    ``p -= lr*wd*p`` is applied after the Adam update."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


def train_step(P: Params, batch: dict, arch, opt_state: dict, truncate_text: bool = True, none_grad: str = "zero"):
    """Full single-rank step on leaf copies of P: losses, grads, HF-AdamW update (in place on P).

    opt_state: {'steps': {name: int}, 'm': {name: Tensor}, 'v': {name: Tensor}}.

    A tensor that took no part in this step's loss -- every `pred_model.*` tensor in a WebVid step (NT = 1,
    model_dist_TVTSv2_ViT_B_16.py:87-90, trainer.py:494) -- follows the reference's loop under its PINNED torch 1.11
    (v2/requirement.txt:142): `self.optimizer.zero_grad()` (trainer.py:477, no argument) leaves a ZERO TENSOR in `.grad`
    once a gradient has existed (set_to_none became the default in torch 2.0), HF AdamW only skips `p.grad is None`, so the
    tensor is updated with g = 0: both moments decay, the weights move by -step_size * m / (sqrt(v) + eps) and by the
    decoupled decay.  A tensor that has NEVER had a gradient is skipped and its per-parameter step count does not advance
    (none arises in the kept configs: loader 0 is YT-Temporal in all three, so the first step reaches every trainable tensor).
    Pinned by tests/golden/alternating_steps.npz (the reference's modules driven through YT / WebVid / YT ... steps).
    none_grad="skip" is modern torch's set_to_none behaviour, kept as the negative control of that test."""
    groups, frozen = param_groups(list(P.keys()), arch)
    leaves = {k: t.detach().clone().requires_grad_(k not in frozen) for k, t in P.items()}
    loss1, loss2, *_ = step_losses(leaves, batch, arch, truncate_text)
    (loss1 + loss2).backward()
    steps = opt_state.setdefault("steps", {})
    grads = {}
    for gi, names in enumerate(groups):
        lr, wd = GROUP_HPARAMS[gi]
        for k in names:
            g = leaves[k].grad
            if g is None:
                if none_grad == "skip" or k not in steps:
                    continue
                g = torch.zeros_like(P[k])
            else:
                grads[k] = g
            m = opt_state.setdefault("m", {}).setdefault(k, torch.zeros_like(P[k]))
            v = opt_state.setdefault("v", {}).setdefault(k, torch.zeros_like(P[k]))
            steps[k] = steps.get(k, 0) + 1
            hf_adamw_step(P[k], g, m, v, steps[k], lr, wd)
    opt_state["step"] = opt_state.get("step", 0) + 1
    return float(loss1.detach()), float(loss2.detach()), grads


# --------------------------------------------------------------------------------------
# validation metrics (v2/model/metric.py:16-126,129-187,285-296; trainer.py:527-635) -- SURVEY.md 8f N1
# --------------------------------------------------------------------------------------

def t2v_ranks(sims, query_masks=None):
    """metric.py:28-66,104-111: rank of each text query's own video in sims [n_text, n_vid], ties broken optimistically
    (the first position of the ground-truth distance in the sorted row); masked queries are dropped."""
    import numpy as np
    sims = np.asarray(sims, dtype=np.float32)
    nq, nv = sims.shape
    q = nq // nv
    dists = -sims
    sorted_d = np.sort(dists, axis=1)
    cols = np.empty(nq, dtype=np.float64)
    for i in range(nq):
        cols[i] = np.where(sorted_d[i] - dists[i, i // q] == 0)[0][0]
    if query_masks is not None:
        cols = cols[np.asarray(query_masks).reshape(-1).astype(bool)]
    return cols


def v2t_ranks(sims, query_masks=None):
    """metric.py:143-187: for each video the best rank among its own captions in sims.T, ties averaged; a missing caption
    gets the distance 1e8 (:160-166) and is skipped as a target (:175-177)."""
    import numpy as np
    d = -np.asarray(sims, dtype=np.float32).T  # [n_vid, n_caps]
    nv, nc = d.shape
    q = nc // nv
    if query_masks is not None:
        d = d.copy()
        d[:, np.logical_not(np.asarray(query_masks).reshape(-1).astype(bool))] = 1e8
    out = np.empty(nv, dtype=np.float64)
    for i in range(nv):
        sd = np.sort(d[i])
        out[i] = min((np.where(sd - d[i, j] == 0)[0].mean() for j in range(i * q, (i + 1) * q) if d[i, j] != 1e8),
                     default=np.inf)
    return out


def cols2metrics(cols, num_queries=None):
    """metric.py:285-296."""
    import numpy as np
    cols = np.asarray(cols, dtype=np.float64)
    n = len(cols) if num_queries is None else num_queries
    m = {"R1": 100 * float(np.sum(cols == 0)) / n, "R5": 100 * float(np.sum(cols < 5)) / n,
         "R10": 100 * float(np.sum(cols < 10)) / n, "R50": 100 * float(np.sum(cols < 50)) / n,
         "MedR": float(np.median(cols) + 1), "MeanR": float(np.mean(cols) + 1)}
    stats = [m["R1"], m["R5"], m["R10"]]
    m["geometric_mean_R1-R5-R10"] = float(np.exp(np.mean(np.log(stats)))) if min(stats) > 0 else 0.0
    return m


def sorting_accuracy(pred_argmax, labels):
    """trainer.py:575-583: a sample counts when ALL of its NT predicted positions are right -> (hits, samples)."""
    import numpy as np
    pred_argmax, labels = np.asarray(pred_argmax), np.asarray(labels)
    return int(np.all(pred_argmax == labels, axis=1).sum()), int(pred_argmax.shape[0])


def step_flops_per_pair(arch, T: int, caption_len: int = 32, NT: int = 4) -> Tuple[float, float]:
    """Algorithmic matmul FLOPs per video-text pair (fwd, bwd) -- SURVEY.md 8d formula."""
    p, W, E, Wt = arch["patch"], arch["width"], arch["embed"], arch["text_width"]
    n = n_keep(arch)
    S = 1 + T * n
    layers, Lt, L = arch["layers"], arch["text_layers"], caption_len
    So = (S if arch["tail"] == "all_tokens" else S - 1) + NT
    patch = 2 * T * n * 3 * p * p * W
    text_gemm = NT * Lt * L * 24 * Wt * Wt
    fwd = (patch + layers * S * 32 * W * W + layers * (4 * n * T * (T + 1) * W + 4 * T * n * (n + 1) * W + 8 * S * W)
           + 2 * S * W * E + text_gemm + NT * Lt * 4 * L * L * Wt + NT * 2 * Wt * E
           + 2 * So * 24 * E * E + 8 * So * So * E + 32 * E)
    frozen = arch["text_tune_from"]
    bwd = 2 * fwd - patch - (frozen / Lt) * text_gemm
    return float(fwd), float(bwd)


# ---------------------------------------------------------------------------------------------------------------------
# tube mask (N3, input pipeline): the reference draws np.random.shuffle(arange(ppf))[:n_keep] per sample in the dataset
# worker (v2/data_loader/YTTemporal_dataset.py:207-213).  A worker's Mersenne-Twister stream is not part of any contract,
# so the device generator is pinned on (a) this exact integer restatement of ITS counter-based draw and (b) the
# reference's distributional contract: an unsorted n_keep-prefix of a uniformly random permutation of range(ppf).
# ---------------------------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def tube_mask(seed, first_sample, B, ppf, n_keep):
    """int32 [B, n_keep]: patch indices of sample first_sample + b ordered by their 54-bit random keys (ties cannot occur:
    the patch index is the low 10 bits of the sorted word)."""
    import numpy as _np
    out = _np.zeros((B, n_keep), dtype=_np.int32)
    for b in range(B):
        base = _splitmix64(((seed & _M64) + _splitmix64(((first_sample & _M64) + b) & _M64)) & _M64)
        words = sorted(((_splitmix64((base + i) & _M64) & ~0x3FF) | i) for i in range(ppf))
        out[b] = [w & 0x3FF for w in words[:n_keep]]
    return out
