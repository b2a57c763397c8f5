"""CPU oracle for the v1 TVTS pretrain step -- TEST INFRASTRUCTURE ONLY.

fp32 torch-CPU restatement of the reference's v1 path: Conv3d tubelet patch embedding + per-tube masking + joint
space-time attention ViT (v1/model/video_encoder.py:78-217), the DistilBERT text tower the reference takes from Hugging
Face (v1/model/model_dist_TVTS.py:34,131-141), the ReLU+Linear / Linear projections (:65-72), the sorting head (the same
SortTransformer as v2, v1/model/sort_transformer.py) and the step's losses (v1/trainer/trainer.py:137-152).  Only
``tests/`` may import it.

Parity status: PINNED against the real v1 classes (``TVTS``, ``VisionTransformer``) imported in the build container by
tests/golden/make_golden_v1.py (fixtures tests/golden/v1_*.npz).  The text tower is third-party code that is absent from
the reference tree: ``transformers.AutoModel.from_pretrained('distilbert-base-uncased')``, transformers pinned at 4.10.2
in v1's environment; the fixture runs the DistilBertModel class of the transformers 5.15.0 installed in this image with
seeded random weights (no pretrained weights here) -- its forward is the published DistilBERT algorithm restated in
``distilbert``.  Dropout (0.1 in the text tower while training, model_dist_TVTS.py:35) is NOT modelled: the fixture and
this file run it at p = 0.  The optimizer is the same unpinned HF AdamW as in oracle/tvts_oracle.py, one parameter group
(v1/configs/dist-yt-pt.json: lr 1e-4, weight_decay 0).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .tvts_oracle import (_key_seed, gelu_erf, layer_norm, linear, norm_softmax_loss, sim_matrix, sort_head,  # noqa: F401
                          sorting_ce, _softmax_attend)

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# v1/model/model_dist_TVTS.py:38-47 (base_patch16_224), v1/configs/dist-yt-pt.json (num_frames 16 -> 8 tubes in the
# temporal table; the loader feeds 4 frames = 2 tubes), v1/data_loader/YTTemporal_dataset.py:68 (mask ratio 0.75)
ARCH = dict(name="v1", image=224, patch=16, tubelet=2, width=768, heads=12, layers=12, num_frames=16, mask_ratio=0.75,
            text_width=768, text_heads=12, text_layers=6, text_ffn=3072, vocab=30522, max_pos=512, embed=256,
            sort_width=768, sort_heads=12, sort_depth=2, n_trans=4)


def tiny_arch(**over) -> dict:
    a = dict(name="v1_tiny", image=64, patch=16, tubelet=2, width=128, heads=2, layers=2, num_frames=8, mask_ratio=0.5,
             text_width=128, text_heads=2, text_layers=2, text_ffn=256, vocab=1000, max_pos=40, embed=64,
             sort_width=128, sort_heads=2, sort_depth=2, n_trans=4)
    a.update(over)
    return a


def patches_per_frame(a) -> int:
    return (a["image"] // a["patch"]) ** 2


def n_keep(a) -> int:
    return int(patches_per_frame(a) * (1 - a["mask_ratio"]))  # YTTemporal_dataset.py:209


def param_shapes(a) -> "Dict[str, Tuple[int, ...]]":
    """State-dict keys of v1 ``TVTS`` in registration order: text_model (HF DistilBertModel), video_model, txt_proj,
    vid_proj, pred_model (model_dist_TVTS.py:34,58,62-76)."""
    out: Dict[str, Tuple[int, ...]] = {}
    Wt, F_, W, E, Ws = a["text_width"], a["text_ffn"], a["width"], a["embed"], a["sort_width"]
    out["text_model.embeddings.word_embeddings.weight"] = (a["vocab"], Wt)
    out["text_model.embeddings.position_embeddings.weight"] = (a["max_pos"], Wt)
    out["text_model.embeddings.LayerNorm.weight"] = (Wt,)
    out["text_model.embeddings.LayerNorm.bias"] = (Wt,)
    for i in range(a["text_layers"]):
        p = f"text_model.transformer.layer.{i}."
        for lin in ("q_lin", "k_lin", "v_lin", "out_lin"):
            out[p + f"attention.{lin}.weight"] = (Wt, Wt)
            out[p + f"attention.{lin}.bias"] = (Wt,)
        out[p + "sa_layer_norm.weight"] = (Wt,)
        out[p + "sa_layer_norm.bias"] = (Wt,)
        out[p + "ffn.lin1.weight"] = (F_, Wt)
        out[p + "ffn.lin1.bias"] = (F_,)
        out[p + "ffn.lin2.weight"] = (Wt, F_)
        out[p + "ffn.lin2.bias"] = (Wt,)
        out[p + "output_layer_norm.weight"] = (Wt,)
        out[p + "output_layer_norm.bias"] = (Wt,)
    ppf, tubes = patches_per_frame(a), a["num_frames"] // a["tubelet"]
    out["video_model.cls_token"] = (1, 1, W)
    out["video_model.pos_embed"] = (1, ppf + 1, W)
    out["video_model.temporal_embed"] = (1, tubes, W)
    out["video_model.patch_embed.proj.weight"] = (W, 3, a["tubelet"], a["patch"], a["patch"])
    out["video_model.patch_embed.proj.bias"] = (W,)
    for i in range(a["layers"]):
        p = f"video_model.blocks.{i}."
        out[p + "norm1.weight"] = (W,); out[p + "norm1.bias"] = (W,)
        out[p + "attn.qkv.weight"] = (3 * W, W); out[p + "attn.qkv.bias"] = (3 * W,)
        out[p + "attn.proj.weight"] = (W, W); out[p + "attn.proj.bias"] = (W,)
        out[p + "norm2.weight"] = (W,); out[p + "norm2.bias"] = (W,)
        out[p + "mlp.fc1.weight"] = (4 * W, W); out[p + "mlp.fc1.bias"] = (4 * W,)
        out[p + "mlp.fc2.weight"] = (W, 4 * W); out[p + "mlp.fc2.bias"] = (W,)
    out["video_model.norm.weight"] = (W,); out["video_model.norm.bias"] = (W,)
    out["txt_proj.1.weight"] = (E, Wt); out["txt_proj.1.bias"] = (E,)
    out["vid_proj.0.weight"] = (E, W); out["vid_proj.0.bias"] = (E,)
    out["pred_model.type_embed"] = (1, 2, Ws)
    for i in range(a["sort_depth"]):
        p = f"pred_model.blocks.{i}."
        out[p + "norm1.weight"] = (Ws,); out[p + "norm1.bias"] = (Ws,)
        out[p + "attn.qkv.weight"] = (3 * Ws, Ws); out[p + "attn.qkv.bias"] = (3 * Ws,)
        out[p + "attn.proj.weight"] = (Ws, Ws); out[p + "attn.proj.bias"] = (Ws,)
        out[p + "norm2.weight"] = (Ws,); out[p + "norm2.bias"] = (Ws,)
        out[p + "mlp.fc1.weight"] = (4 * Ws, Ws); out[p + "mlp.fc1.bias"] = (4 * Ws,)
        out[p + "mlp.fc2.weight"] = (Ws, 4 * Ws); out[p + "mlp.fc2.bias"] = (Ws,)
    out["pred_model.norm.weight"] = (Ws,); out["pred_model.norm.bias"] = (Ws,)
    out["pred_model.head.weight"] = (a["n_trans"], Ws); out["pred_model.head.bias"] = (a["n_trans"],)
    return out


def synth_params(a, seed: int = 0) -> Params:
    """Deterministic per-key synthetic parameters at a trained checkpoint's order of magnitude."""
    out: Params = {}
    for name, shape in param_shapes(a).items():
        g = torch.Generator().manual_seed(_key_seed(seed, name))
        t = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = any(s in name for s in ("norm", "LayerNorm"))
        if is_norm and leaf == "weight":
            t = 1.0 + 0.05 * t
        elif leaf == "bias":
            t = 0.02 * t
        elif "embeddings" in name:
            t = 0.02 * t
        elif name in ("video_model.cls_token", "video_model.pos_embed", "video_model.temporal_embed", "pred_model.type_embed"):
            t = 0.02 * t
        elif name == "video_model.patch_embed.proj.weight":
            t = t * (3 * a["tubelet"] * a["patch"] ** 2) ** -0.5
        else:
            t = t * (shape[-1] ** -0.5) * 0.7
        out[name] = t
    return out


def synth_batch(a, B: int, T: int, seed: int = 0, n_trans: Optional[int] = None, caption_len: int = 12):
    """v1 batch dict (v1/data_loader/YTTemporal_dataset.py:200-240, trainer.py:121-131): video fp32 [B,T,3,H,W];
    text = tokenizer output {'input_ids','attention_mask'} [NT*B, L] padded to the longest caption of the batch
    (clip-major rows), [CLS] = 101 first, [SEP] = 102 last, pad 0; keep_ind int64 [B, n_tubes, n_keep], an unsorted prefix
    of a permutation per TUBE (:211-215); label [B,4] = arange(4)."""
    NT = a["n_trans"] if n_trans is None else n_trans
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, T, 3, a["image"], a["image"], generator=g, dtype=torch.float32)
    N, L = NT * B, caption_len
    lens = torch.randint(max(3, L // 2), L + 1, (N,), generator=g)
    lens[0] = L
    ids = torch.zeros(N, L, dtype=torch.int64)
    mask = torch.zeros(N, L, dtype=torch.int64)
    cls_id, sep_id = (101, 102) if a["vocab"] > 1000 else (a["vocab"] - 2, a["vocab"] - 1)
    for r in range(N):
        n = int(lens[r])
        ids[r, 0] = cls_id
        ids[r, 1:n - 1] = torch.randint(1, min(a["vocab"], 30000) - 2, (n - 2,), generator=g)
        ids[r, n - 1] = sep_id
        mask[r, :n] = 1
    ppf, nk, tubes = patches_per_frame(a), n_keep(a), T // a["tubelet"]
    keep = torch.stack([torch.stack([torch.randperm(ppf, generator=g)[:nk] for _ in range(tubes)]) for _ in range(B)])
    batch = {"video": video, "text": {"input_ids": ids, "attention_mask": mask}, "keep_ind": keep.to(torch.int64)}
    if NT == a["n_trans"]:
        batch["label"] = torch.arange(a["n_trans"]).repeat(B, 1)
    return batch


# ------------------------------------------------------------------------------------------------ dropout masks
# The build's counter-based generator, restated (tvts_amd/csrc/attention.hip::drop_keep, embed.hip::dropout_rows_kernel):
# element i of call site k at seed s is KEPT iff the upper 32 bits of splitmix64(s + k * SITE_STRIDE + i * PHI) >= p * 2^32.
# (transformers' nn.Dropout draws from torch's generator instead: the masks themselves are the build's own choice, what is pinned
# against the reference is WHERE dropout acts and how it scales -- tests/golden/make_golden_v1.py runs the real DistilBertModel
# with torch.nn.functional.dropout replaced by this very function.)
DROP_SITE_STRIDE = 0x632BE59BD9B4E019
DROP_STEP_STRIDE = 0x51ED270B7F4A7C15
_M64 = (1 << 64) - 1


def drop_mask(seed: int, site: int, shape, p: float) -> Tensor:
    """float32 tensor of `shape`: 1 / (1 - p) where kept, 0 where dropped"""
    import numpy as np
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        base = np.uint64((seed + site * DROP_SITE_STRIDE) & _M64)
        z = base + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    keep = (z >> np.uint64(32)) >= np.uint64(int(p * 4294967296.0))
    return torch.from_numpy((keep.astype(np.float32) / np.float32(1.0 - p)).reshape(shape))


def trim_text(text: dict) -> dict:
    """the tokenizer's right-padded batch cut to its longest caption (what the build's prepare_batch does; the network function
    is unchanged -- padded keys are masked -- but the dropout masks are indexed by the padded length)"""
    L = int(text["attention_mask"].sum(-1).max())
    return {"input_ids": text["input_ids"][:, :L], "attention_mask": text["attention_mask"][:, :L]}


# ------------------------------------------------------------------------------------------------ text tower
def distilbert(P: Params, ids: Tensor, mask: Tensor, a, drop=None) -> Tensor:
    """DistilBertModel(input_ids, attention_mask).last_hidden_state (transformers modeling_distilbert: Embeddings,
    TransformerBlock with MultiHeadSelfAttention + FFN, POST-LayerNorm, eps 1e-12, erf-GELU; padded keys masked).
    drop = dict(p=0.1, seed=int): training mode -- dropout(LayerNorm(embeddings)) [site 0], dropout(softmax(scores)) [site 1 + 2i],
    dropout(lin2(gelu(lin1(x)))) [site 2 + 2i]; ids / mask must then be trimmed to the longest caption (trim_text)."""
    N, L = ids.shape
    h, Wt = a["text_heads"], a["text_width"]
    dh = Wt // h
    dp = float(drop["p"]) if drop else 0.0
    x = P["text_model.embeddings.word_embeddings.weight"][ids] + P["text_model.embeddings.position_embeddings.weight"][:L]
    x = layer_norm(x, P["text_model.embeddings.LayerNorm.weight"], P["text_model.embeddings.LayerNorm.bias"], 1e-12)
    if dp > 0:
        x = x * drop_mask(drop["seed"], 0, (N, L, Wt), dp)
    key_mask = (mask == 0)[:, None, None, :]  # [N,1,1,L]
    for i in range(a["text_layers"]):
        p = f"text_model.transformer.layer.{i}."
        q = linear(x, P[p + "attention.q_lin.weight"], P[p + "attention.q_lin.bias"]).reshape(N, L, h, dh).permute(0, 2, 1, 3)
        k = linear(x, P[p + "attention.k_lin.weight"], P[p + "attention.k_lin.bias"]).reshape(N, L, h, dh).permute(0, 2, 1, 3)
        v = linear(x, P[p + "attention.v_lin.weight"], P[p + "attention.v_lin.bias"]).reshape(N, L, h, dh).permute(0, 2, 1, 3)
        s = (q * dh ** -0.5) @ k.transpose(-1, -2)
        s = s.masked_fill(key_mask, torch.finfo(s.dtype).min)
        w = torch.softmax(s, dim=-1)
        if dp > 0:
            w = w * drop_mask(drop["seed"], 1 + 2 * i, (N, h, L, L), dp)
        o = (w @ v).permute(0, 2, 1, 3).reshape(N, L, Wt)
        sa = linear(o, P[p + "attention.out_lin.weight"], P[p + "attention.out_lin.bias"])
        x = layer_norm(sa + x, P[p + "sa_layer_norm.weight"], P[p + "sa_layer_norm.bias"], 1e-12)
        f = linear(gelu_erf(linear(x, P[p + "ffn.lin1.weight"], P[p + "ffn.lin1.bias"])), P[p + "ffn.lin2.weight"],
                   P[p + "ffn.lin2.bias"])
        if dp > 0:
            f = f * drop_mask(drop["seed"], 2 + 2 * i, (N, L, Wt), dp)
        x = layer_norm(f + x, P[p + "output_layer_norm.weight"], P[p + "output_layer_norm.bias"], 1e-12)
    return x


def compute_text(P: Params, text: dict, a, drop=None) -> Tuple[Tensor, Tensor]:
    """model_dist_TVTS.py:131-141: [CLS] row of the last hidden state, then txt_proj = Linear(ReLU(.)) (:65-68)."""
    if drop:
        text = trim_text(text)
    before = distilbert(P, text["input_ids"], text["attention_mask"], a, drop)[:, 0, :]
    return before, linear(torch.relu(before), P["txt_proj.1.weight"], P["txt_proj.1.bias"])


# ------------------------------------------------------------------------------------------------ video tower
def video_tokens(P: Params, video: Tensor, keep_ind: Tensor, a) -> Tensor:
    """VisionTransformer.forward_features up to the masking (video_encoder.py:178-207): Conv3d tubelet embedding of ALL
    patches as an explicit im2col product, + cat(pos[0], pos[1:] tiled over tubes + temporal[t] repeated over patches),
    then the kept patches of every tube (keep_ind[b, t] differs per tube)."""
    B, T = video.shape[:2]
    p, tb, W, g = a["patch"], a["tubelet"], a["width"], a["image"] // a["patch"]
    tubes = T // tb
    x = video.reshape(B, tubes, tb, 3, g, p, g, p).permute(0, 1, 4, 6, 3, 2, 5, 7)  # b tube gy gx | c t py px
    cols = x.reshape(B, tubes * g * g, 3 * tb * p * p)
    tok = cols @ P["video_model.patch_embed.proj.weight"].reshape(W, -1).t() + P["video_model.patch_embed.proj.bias"]
    pos = P["video_model.pos_embed"][0]
    tok = tok.reshape(B, tubes, g * g, W) + pos[1:][None, None] + P["video_model.temporal_embed"][0, :tubes][None, :, None]
    keep = keep_ind[:, :tubes].to(torch.int64)
    kept = torch.gather(tok, 2, keep[..., None].expand(-1, -1, -1, W)).reshape(B, -1, W)
    cls = (P["video_model.cls_token"][0] + pos[:1]).expand(B, -1, -1)
    return torch.cat([cls, kept], 1)


def vit(P: Params, x: Tensor, a) -> Tensor:
    """pre-LN blocks with JOINT attention over all kept tokens of the clip (video_encoder.py:34-75,209-214), eps 1e-6."""
    B, S, W = x.shape
    h = a["heads"]
    dh = W // h
    for i in range(a["layers"]):
        p = f"video_model.blocks.{i}."
        y = layer_norm(x, P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-6)
        qkv = linear(y, P[p + "attn.qkv.weight"], P[p + "attn.qkv.bias"]).reshape(B, S, 3, h, dh)
        o = _softmax_attend(qkv[:, :, 0].permute(0, 2, 1, 3) * dh ** -0.5, qkv[:, :, 1].permute(0, 2, 1, 3),
                            qkv[:, :, 2].permute(0, 2, 1, 3))
        x = x + linear(o.permute(0, 2, 1, 3).reshape(B, S, W), P[p + "attn.proj.weight"], P[p + "attn.proj.bias"])
        y = layer_norm(x, P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-6)
        x = x + linear(gelu_erf(linear(y, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"])), P[p + "mlp.fc2.weight"],
                       P[p + "mlp.fc2.bias"])
    return layer_norm(x, P["video_model.norm.weight"], P["video_model.norm.bias"], 1e-6)


def compute_video(P: Params, video: Tensor, keep_ind: Tensor, a) -> Tuple[Tensor, Tensor]:
    """model_dist_TVTS.py:143-147: all normed tokens (sort head input) and vid_proj of the CLS token."""
    before = vit(P, video_tokens(P, video, keep_ind, a), a)
    return before, linear(before[:, 0], P["vid_proj.0.weight"], P["vid_proj.0.bias"])


def model_forward(P: Params, batch: dict, a, drop=None):
    """TVTS.forward (model_dist_TVTS.py:93-123) -> (text_emb [B,E], video_emb [B,E], pred [B,NT,4] | None).
    drop: the text tower's training-mode dropout (distilbert)."""
    B = batch["video"].shape[0]
    before, emb = compute_text(P, batch["text"], a, drop)
    NT = before.shape[0] // B
    text_before = before.reshape(NT, B, -1).detach().permute(1, 0, 2)
    text_emb = emb.reshape(NT, B, -1).mean(0)
    vbefore, video_emb = compute_video(P, batch["video"], batch["keep_ind"], a)
    sa = dict(embed=a["sort_width"], sort_heads=a["sort_heads"], sort_depth=a["sort_depth"])
    pred = sort_head(P, text_before, vbefore, sa) if NT != 1 else None
    return text_emb, video_emb, pred


def step_losses(P: Params, batch: dict, a, drop=None):
    """v1/trainer/trainer.py:137-152: sim_matrix(video, text) -> NormSoftmaxLoss; 2 x CE on the predicted order."""
    te, ve, pred = model_forward(P, batch, a, drop)
    loss1 = norm_softmax_loss(sim_matrix(ve, te))
    loss2 = sorting_ce(pred, batch["label"]) if pred is not None else torch.zeros(())
    return loss1, loss2, te, ve, pred
