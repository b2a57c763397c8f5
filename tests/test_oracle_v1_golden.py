"""The v1 oracle (oracle/tvts_v1_oracle.py) against fixtures produced by the real v1 classes (tests/golden/make_golden_v1.py
ran /root/reference/v1 in the build container): <= 1e-5 on activations / losses, <= 1e-4 on gradients."""
import numpy as np
import pytest
import torch

from oracle import tvts_v1_oracle as V


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _check(f, a, B, T, full=False, drop=None):
    P = {k: v.clone().requires_grad_(True) for k, v in V.synth_params(a, seed=int(f["seed"])).items()}
    assert [str(s) for s in f["param_names"]] == list(P.keys())  # state-dict keys and order of the real class
    batch = V.synth_batch(a, B=B, T=T, seed=int(f["batch_seed"]), caption_len=int(f["caption_len"]))
    l1, l2, te, ve, pred = V.step_losses(P, batch, a, drop)
    assert rel(te, f["te"]) < 1e-5 and rel(ve, f["ve"]) < 1e-5 and rel(pred, f["pred"]) < 1e-5
    assert abs(float(l1) - float(f["loss1"])) < 1e-5 * max(1, abs(float(f["loss1"])))
    assert abs(float(l2) - float(f["loss2"])) < 1e-5 * max(1, abs(float(f["loss2"])))
    (l1 + l2).backward()
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    tot = 0.0
    for k, p in P.items():
        if p.grad is None:
            assert k not in ref or ref[k] == 0.0, k
            continue
        tot += float(p.grad.double().norm()) ** 2
        if ref[k] > 1e-6 * float(f["grad_norm"]):
            assert abs(float(p.grad.norm()) - ref[k]) < 1e-4 * ref[k] + 1e-9, (k, float(p.grad.norm()), ref[k])
    assert abs(tot ** 0.5 - float(f["grad_norm"])) < 1e-4 * float(f["grad_norm"])
    return P


def test_v1_tiny_against_reference(golden):
    f = golden("v1_tiny")
    a = V.tiny_arch()
    P = _check(f, a, int(f["B"]), int(f["T"]))
    sel = {"g_conv": ("video_model.patch_embed.proj.weight", (slice(0, 4),)), "g_pos": ("video_model.pos_embed", ()),
           "g_temporal": ("video_model.temporal_embed", ()), "g_cls": ("video_model.cls_token", ()),
           "g_qkv1": ("video_model.blocks.1.attn.qkv.weight", (slice(0, 16),)),
           "g_word": ("text_model.embeddings.word_embeddings.weight", ()),
           "g_posemb": ("text_model.embeddings.position_embeddings.weight", ()),
           "g_qlin0": ("text_model.transformer.layer.0.attention.q_lin.weight", ()),
           "g_lin2": ("text_model.transformer.layer.1.ffn.lin2.weight", ()), "g_txtproj": ("txt_proj.1.weight", ()),
           "g_vidproj": ("vid_proj.0.weight", ()), "g_head": ("pred_model.head.weight", ())}
    for key, (name, idx) in sel.items():
        g = P[name].grad[idx] if idx else P[name].grad
        assert rel(g, f[key]) < 1e-4, (key, rel(g, f[key]))


def test_v1_training_mode_dropout_against_reference(golden):
    """The reference trains with the text tower in train() mode (v1/model/model_dist_TVTS.py:33-34): DistilBERT's three dropouts
    at p = 0.1.  The fixture ran the REAL transformers DistilBertModel (dropout / attention_dropout 0.1, train()) inside the
    reference's TVTS.forward with nn.functional.dropout replaced by the counter-based mask generator: same sites, same scaling,
    same masks here -> the oracle's training-mode forward, losses and gradients to fp32 round-off."""
    f = golden("v1_tiny_dropout")
    a = V.tiny_arch()
    assert int(f["n_dropout_calls"]) == 1 + 2 * a["text_layers"]
    drop = dict(p=float(f["p"]), seed=int(f["drop_seed"]))
    P = _check(f, a, int(f["B"]), int(f["T"]), drop=drop)
    sel = {"g_word": "text_model.embeddings.word_embeddings.weight", "g_qlin0": "text_model.transformer.layer.0.attention.q_lin.weight",
           "g_vlin1": "text_model.transformer.layer.1.attention.v_lin.weight", "g_lin2": "text_model.transformer.layer.1.ffn.lin2.weight",
           "g_lin1": "text_model.transformer.layer.0.ffn.lin1.weight", "g_txtproj": "txt_proj.1.weight", "g_head": "pred_model.head.weight"}
    for key, name in sel.items():
        assert rel(P[name].grad, f[key]) < 1e-4, (key, rel(P[name].grad, f[key]))
    # and it is not the p = 0 model: the same parameters and batch without dropout give a different text embedding
    batch = V.synth_batch(a, B=int(f["B"]), T=int(f["T"]), seed=int(f["batch_seed"]), caption_len=int(f["caption_len"]))
    with torch.no_grad():
        te0 = V.model_forward({k: v.detach() for k, v in P.items()}, batch, a)[0]
    assert rel(te0, f["te"]) > 1e-2
    # keep rate of the generator
    m = V.drop_mask(123, 5, (200000,), 0.1)
    assert abs(float((m > 0).float().mean()) - 0.9) < 3e-3 and float(m.max()) == pytest.approx(1 / 0.9)


def test_v1_tiny_single_caption_batch(golden):
    """one caption per video (WebVid-style): no sorting head, pred None (model_dist_TVTS.py:113-116)"""
    f = golden("v1_tiny_nt1")
    a = V.tiny_arch()
    P = V.synth_params(a, seed=int(f["seed"]))
    batch = V.synth_batch(a, B=3, T=4, seed=int(f["batch_seed"]), n_trans=1, caption_len=9)
    l1, l2, te, ve, pred = V.step_losses(P, batch, a)
    assert pred is None and float(l2) == 0.0
    assert rel(te, f["te"]) < 1e-5 and rel(ve, f["ve"]) < 1e-5 and abs(float(l1) - float(f["loss1"])) < 1e-5


def test_v1_full_size_against_reference(golden):
    """the real TVTS class: DistilBERT-base + tubelet ViT-B/16 + sorting head at B=2, 4 frames, mask 0.75"""
    f = golden("v1_full")
    torch.set_num_threads(8)
    P = _check(f, V.ARCH, int(f["B"]), int(f["T"]), full=True)
    for key, (name, idx) in {"g_conv": ("video_model.patch_embed.proj.weight", (slice(0, 2),)),
                             "g_temporal": ("video_model.temporal_embed", (slice(None), slice(None), slice(0, 32))),
                             "g_qkv11": ("video_model.blocks.11.attn.qkv.weight", (slice(0, 8), slice(0, 32))),
                             "g_qlin5": ("text_model.transformer.layer.5.attention.q_lin.weight", (slice(0, 8), slice(0, 32))),
                             "g_txtproj": ("txt_proj.1.weight", (slice(0, 8), slice(0, 32))),
                             "g_vidproj": ("vid_proj.0.weight", (slice(0, 8), slice(0, 32))),
                             "g_head": ("pred_model.head.weight", (slice(None), slice(0, 64)))}.items():
        assert rel(P[name].grad[idx], f[key]) < 2e-4, (key, rel(P[name].grad[idx], f[key]))
