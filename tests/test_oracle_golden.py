"""Pins the CPU oracle (oracle/tvts_oracle.py) against the fixtures produced by running the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import tvts_oracle as O

RTOL = 1e-5   # fp32 restatement vs reference: <= 1e-5 rel on activations / losses
GTOL = 1e-4   # <= 1e-4 rel on gradients / grad-norm


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64) if not isinstance(b, torch.Tensor) else b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def leaves(P):
    return {k: v.clone().requires_grad_(True) for k, v in P.items()}


def test_block(golden):
    f = golden("block_tiny")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    x = torch.tensor(f["x"], requires_grad=True)
    pre = "video_model.transformer.resblocks.1."
    y = O.st_block(x, P, pre, arch, int(f["T"]), int(f["n"]))
    assert relerr(f["y"], y.detach()) < RTOL
    (y * torch.tensor(f["r"])).sum().backward()
    assert relerr(f["gx"], x.grad) < GTOL
    for k in f.files:
        if k.startswith("g_") and k != "gx":
            assert relerr(f[k], P[pre + k[2:]].grad) < GTOL, k


def test_vit(golden):
    f = golden("vit_tiny")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    b = O.synth_batch(arch, B=int(f["B"]), T=int(f["T"]), seed=int(f["batch_seed"]))
    _, out = O.video_tower(P, b["video"], b["keep_ind"], arch)
    assert relerr(f["out"], out.detach()) < RTOL
    (out * torch.tensor(f["r"])).sum().backward()
    for k in f.files:
        if k.startswith("g_"):
            assert relerr(f[k], P["video_model." + k[2:]].grad) < GTOL, k


@pytest.mark.parametrize("truncate", [True, False])
def test_text(golden, truncate):
    f = golden("text_tiny")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    emb = O.text_tower(P, torch.tensor(f["ids"]), arch, truncate=truncate)
    assert relerr(f["emb"], emb.detach()) < RTOL
    (emb * torch.tensor(f["r"])).sum().backward()
    assert relerr(f["g_tok"], P["text_token_embedding.weight"].grad) < GTOL
    assert relerr(f["g_pos"], P["text_positional_embedding"].grad) < GTOL
    assert relerr(f["g_proj"], P["text_projection"].grad) < GTOL
    assert relerr(f["g_qkv0"], P["text_model.resblocks.0.attn.in_proj_weight"].grad) < GTOL
    assert relerr(f["g_lnf"], P["text_ln_final.weight"].grad) < GTOL


def test_sort(golden):
    f = golden("sort_tiny")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    tok = torch.tensor(f["tok"], requires_grad=True)
    out = O.sort_head(P, torch.tensor(f["text"]), tok, arch)
    assert relerr(f["out"], out.detach()) < RTOL
    (out * torch.tensor(f["r"])).sum().backward()
    assert relerr(f["g_tok"], tok.grad) < GTOL
    assert relerr(f["g_type"], P["pred_model.type_embed"].grad) < GTOL
    assert relerr(f["g_head"], P["pred_model.head.weight"].grad) < GTOL
    assert relerr(f["g_qkv1"], P["pred_model.blocks.1.attn.qkv.weight"].grad) < GTOL
    assert relerr(f["g_fc1b"], P["pred_model.blocks.0.mlp.fc1.bias"].grad) < GTOL


def _check_model(f, arch, P, batch):
    l1, l2, te, ve, pred = O.step_losses(P, batch, arch)
    assert relerr(f["te"], te.detach()) < RTOL
    assert relerr(f["ve"], ve.detach()) < RTOL
    assert relerr(f["pred"], pred.detach()) < 10 * RTOL
    assert abs(float(l1) - float(f["loss1"])) < 1e-5 * max(1.0, abs(float(f["loss1"])))
    assert abs(float(l2) - float(f["loss2"])) < 1e-5 * max(1.0, abs(float(f["loss2"])))
    (l1 + l2).backward()
    gn = {k: float(v.grad.norm()) for k, v in P.items() if v.grad is not None}
    total = sum(v * v for v in gn.values()) ** 0.5
    assert abs(total - float(f["grad_norm"])) < GTOL * float(f["grad_norm"])
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    for k, v in ref.items():
        assert abs(gn[k] - float(v)) <= 5e-4 * float(v) + 1e-7 * float(f["grad_norm"]), (k, gn[k], float(v))
    return gn


def test_model_tiny(golden):
    f = golden("model_tiny")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    b = O.synth_batch(arch, B=int(f["B"]), T=int(f["T"]), seed=int(f["batch_seed"]), caption_len=int(f["caption_len"]))
    _check_model(f, arch, P, b)
    for k in f.files:
        if k.startswith("g_") and k not in ("gn_names", "gn_vals", "grad_norm"):
            assert relerr(f[k], P[k[2:]].grad) < GTOL, k


def h_tiny_arch():
    # same dict as tests/golden/make_golden.py:h_tiny_arch
    return O.tiny_arch(name="H_14", image=28, patch=7, width=80, heads=2, layers=2, embed=32, text_width=32,
                       text_heads=2, text_layers=3, text_tune_from=1, act="gelu", tail="pooled_and_patches",
                       block_order="openclip", mask_ratio=0.7)


def test_model_h_tiny(golden):
    """H/14 structure (video_encoder_ViT_H_14.py + OpenCLIP text blocks, wired by TVTSv2_H_14.forward):
    state-dict registration order, erf-GELU, ln_post on CLS only, CLS-less patch tokens into the sort head."""
    f = golden("model_h_tiny")
    arch = h_tiny_arch()
    assert [str(s) for s in f["names"]] == list(O.param_shapes(arch).keys())
    full = [str(s) for s in f["names"]]
    big = list(O.param_shapes(O.ARCHS["H_14"]).keys())
    # the full-size table is the same sequence with more layers: block-local order must agree
    blk = lambda names, pre: [n[len(pre):] for n in names if n.startswith(pre)]  # noqa: E731
    for pre in ("video_model.transformer.resblocks.1.", "text_model.resblocks.2."):
        assert blk(full, pre) == blk(big, pre)
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    b = O.synth_batch(arch, B=int(f["B"]), T=int(f["T"]), seed=int(f["batch_seed"]), caption_len=int(f["caption_len"]))
    _check_model(f, arch, P, b)
    for k in f.files:
        if k.startswith("g_") and k not in ("gn_names", "gn_vals", "grad_norm"):
            assert relerr(f[k], P[k[2:]].grad) < GTOL, k


def test_model_tiny_nt1(golden):
    f = golden("model_tiny_nt1")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    b = O.synth_batch(arch, B=3, T=3, seed=int(f["batch_seed"]), n_trans=1, caption_len=10)
    l1, l2, te, ve, pred = O.step_losses(P, b, arch)
    assert pred is None and float(l2) == 0.0
    assert relerr(f["te"], te.detach()) < RTOL and relerr(f["ve"], ve.detach()) < RTOL
    assert abs(float(l1) - float(f["loss1"])) < 1e-5
    l1.backward()
    assert relerr(f["g_proj"], P["video_model.proj"].grad) < GTOL
    assert relerr(f["g_tproj"], P["text_projection"].grad) < GTOL
    assert P["pred_model.head.weight"].grad is None


def test_model_b32_config1(golden):
    """BASELINE config 1: the real TVTSv2_B_32 class, B=2, T=4 (reference ran at context 77)."""
    f = golden("model_b32_cfg1")
    arch = O.ARCHS["B_32"]
    P = leaves(O.synth_params(arch, seed=0))
    b = O.synth_batch(arch, B=2, T=4, seed=0)
    _check_model(f, arch, P, b)
    sl = {"g_video_proj": ("video_model.proj", (slice(0, 8), slice(0, 16))),
          "g_text_proj": ("text_projection", (slice(0, 8), slice(0, 16))),
          "g_head": ("pred_model.head.weight", (slice(None), slice(None))),
          "g_cfc11": ("video_model.transformer.resblocks.11.mlp.c_fc.weight", (slice(0, 8), slice(0, 16))),
          "g_tqkv0": ("video_model.transformer.resblocks.0.timeattn.qkv.weight", (slice(0, 8), slice(0, 16))),
          "g_temporal": ("video_model.temporal_embedding", (slice(None), slice(0, 16))),
          "g_textqkv10": ("text_model.resblocks.10.attn.in_proj_weight", (slice(0, 8), slice(0, 16)))}
    for k, (name, idx) in sl.items():
        assert relerr(f[k], P[name].grad[idx]) < 2e-4, k


def test_param_groups(golden):
    f = golden("param_groups_b16")
    arch = O.ARCHS["B_16"]
    groups, frozen = O.param_groups(list(O.param_shapes(arch)), arch)
    ref = dict(zip([str(s) for s in f["names"]], [int(g) for g in f["group"]]))
    mine = {n: gi for gi, names in enumerate(groups) for n in names}
    assert mine == ref
    assert sorted(frozen) == sorted(str(s) for s in f["frozen"])
    # App. B #11 spot checks
    assert mine["pred_model.blocks.0.norm1.weight"] == 1 and mine["text_ln_final.weight"] == 3
    assert mine["video_model.positional_embedding"] == 2 and mine["video_model.temporal_embedding"] == 2


def test_ddp2_emulation(golden):
    """World-2 reference run (gloo DDP + AllGather_multi) vs the single-process emulation."""
    f = golden("ddp2_tiny")
    arch = O.tiny_arch()
    P = leaves(O.synth_params(arch, seed=int(f["seed"])))
    bs = [O.synth_batch(arch, B=int(f["B"]), T=int(f["T"]), seed=int(f[f"batch_seed{r}"]),
                        caption_len=int(f["caption_len"])) for r in range(2)]
    total, loss1, l2 = O.multi_rank_step(P, bs, arch)
    assert abs(float(loss1) - float(f["loss1"][0])) < 1e-5 and abs(float(loss1) - float(f["loss1"][1])) < 1e-5
    for r in range(2):
        assert abs(float(l2[r]) - float(f["loss2"][r])) < 1e-5
    total.backward()
    gn = sum(float(v.grad.norm()) ** 2 for v in P.values() if v.grad is not None) ** 0.5
    assert abs(gn - float(f["grad_norm"])) < GTOL * float(f["grad_norm"])
    for k in f.files:
        if k.startswith("g_"):
            assert relerr(f[k], P[k[2:]].grad) < GTOL, k


def test_hf_adamw_restated():
    """Optimizer parity is UNPINNED (transformers.AdamW absent); this checks the restatement's
    algebra against a closed form for two steps."""
    p = torch.tensor([1.0, -2.0]); g = torch.tensor([0.5, 0.25])
    m = torch.zeros(2); v = torch.zeros(2)
    p0 = p.clone()
    O.hf_adamw_step(p, g, m, v, 1, lr=1e-2, wd=0.1)
    m1 = 0.1 * g; v1 = 0.001 * g * g
    upd = 1e-2 * (1 - 0.999) ** 0.5 / (1 - 0.9) * m1 / (v1.sqrt() + 1e-6)
    exp = (p0 - upd) * (1 - 1e-2 * 0.1)
    assert torch.allclose(p, exp, rtol=1e-6, atol=1e-8)


def test_hf_adamw_against_torch_adamw():
    """Independent cross-check of the restated optimizer (transformers.AdamW itself is absent from this image).  HF's update
    lr * sqrt(bc2)/bc1 * m / (sqrt(v) + eps) equals torch.optim.AdamW's lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps') with
    eps' = eps / sqrt(bc2), so torch's implementation pins the moment updates and both bias corrections step by step; the two
    things it cannot pin -- eps outside the bias correction, decay applied to the UPDATED weight -- are the published
    differences of transformers==4.10.2 and enter through that mapping and the (p - u)(1 - lr wd) identity below."""
    torch.manual_seed(0)
    lr, b1, b2, eps = 3e-4, 0.9, 0.999, 1e-6
    p_ref = torch.nn.Parameter(torch.randn(257, dtype=torch.float64))
    opt = torch.optim.AdamW([p_ref], lr=lr, betas=(b1, b2), eps=eps, weight_decay=0.0)
    p = p_ref.detach().clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 8):
        g = torch.randn(257, dtype=torch.float64) * (10.0 ** torch.randint(-4, 2, (257,)).double())
        opt.param_groups[0]["eps"] = eps / (1.0 - b2 ** step) ** 0.5
        p_ref.grad = g.clone()
        opt.step()
        O.hf_adamw_step(p, g, m, v, step, lr=lr, wd=0.0, beta1=b1, beta2=b2, eps=eps)
        st = opt.state[p_ref]
        assert torch.allclose(m, st["exp_avg"], rtol=1e-12, atol=0) and torch.allclose(v, st["exp_avg_sq"], rtol=1e-12, atol=0)
        assert torch.allclose(p, p_ref.detach(), rtol=1e-10, atol=1e-14), step
    # decoupled decay: one more step from the same state with and without it -> p_wd = (p - u) * (1 - lr * wd)
    g = torch.randn(257, dtype=torch.float64)
    pa, pb = p.clone(), p.clone()
    O.hf_adamw_step(pa, g, m.clone(), v.clone(), 8, lr=lr, wd=0.0, beta1=b1, beta2=b2, eps=eps)
    O.hf_adamw_step(pb, g, m.clone(), v.clone(), 8, lr=lr, wd=0.2, beta1=b1, beta2=b2, eps=eps)
    assert torch.allclose(pb, pa * (1.0 - lr * 0.2), rtol=1e-14, atol=0)


def test_flops_formula():
    # BASELINE.md section 3 table
    for name, T, step in (("B_32", 4, 167.5), ("B_32", 8, 313.3), ("B_16", 8, 606.1)):
        f, b = O.step_flops_per_pair(O.ARCHS[name], T)
        assert abs((f + b) / 1e9 - step) / step < 0.01, (name, (f + b) / 1e9)


def test_model_h14_config3(golden):
    """BASELINE config 3 architecture: the real TVTSv2_H_14 class at full size (1.22 G parameters), B=2, T=4."""
    f = golden("model_h14_cfg3")
    arch = O.ARCHS["H_14"]
    P = leaves(O.synth_params(arch, seed=0))
    b = O.synth_batch(arch, B=2, T=4, seed=0)
    _check_model(f, arch, P, b)
    sl = {"g_video_proj": ("video_model.proj", (slice(0, 8), slice(0, 16))),
          "g_text_proj": ("text_projection", (slice(0, 8), slice(0, 16))),
          "g_head": ("pred_model.head.weight", (slice(None), slice(0, 64))),
          "g_lnpost": ("video_model.ln_post.weight", (slice(None),)),
          "g_cfc31": ("video_model.transformer.resblocks.31.mlp.c_fc.weight", (slice(0, 8), slice(0, 16))),
          "g_tqkv0": ("video_model.transformer.resblocks.0.timeattn.qkv.weight", (slice(0, 8), slice(0, 16))),
          "g_temporal": ("video_model.temporal_embedding", (slice(None), slice(0, 16))),
          "g_textqkv20": ("text_model.resblocks.20.attn.in_proj_weight", (slice(0, 8), slice(0, 16)))}
    for k, (name, idx) in sl.items():
        assert relerr(f[k], P[name].grad[idx]) < 5e-4, k
    assert relerr(f["g_conv"], P["video_model.conv1.weight"].grad[:4].reshape(4, -1)) < 5e-4


def test_retrieval_metrics(golden):
    """v2/model/metric.py t2v_metrics / v2t_metrics (executed in the build container) on matrices with and without ties."""
    f = golden("metrics")
    keys = [str(k) for k in f["keys"]]
    for name in ("rand_square", "ties_square", "two_caps", "two_caps_ties"):
        sims = f["sims_" + name]
        for fn, ranks in (("t2v_metrics", O.t2v_ranks), ("v2t_metrics", O.v2t_ranks)):
            got = O.cols2metrics(ranks(sims))
            want = dict(zip(keys, f[f"{fn}_{name}"]))
            for k in keys:
                assert abs(got[k] - want[k]) < 1e-9 * max(1.0, abs(want[k])), (name, fn, k, got[k], want[k])
    # MSRVTT-style missing captions (query_masks)
    sims, mask = f["sims_two_caps"], f["query_mask_two_caps"]
    for fn, ranks in (("t2v_metrics", O.t2v_ranks), ("v2t_metrics", O.v2t_ranks)):
        cols = ranks(sims, mask)
        got = O.cols2metrics(cols, int(mask.sum()) if fn == "t2v_metrics" else None)
        want = dict(zip(keys, f[f"{fn}_two_caps_masked"]))
        for k in keys:
            assert abs(got[k] - want[k]) < 1e-6 * max(1.0, abs(want[k])), ("masked", fn, k, got[k], want[k])  # the reference divides by a float32 mask sum


def test_downstream_b16(golden):
    """The reference's inference-only model copy (v2/downstream/model_TVTSv2_ViT_B_16.py): mask 0, no sort head."""
    f = golden("downstream_b16")
    arch = dict(O.ARCHS["B_16"], mask_ratio=0.0, sort_head=False)
    assert not any(k.startswith("pred_model") for k in O.param_shapes(arch))
    P = O.synth_params(arch, seed=int(f["seed"]))
    b = O.synth_batch(O.ARCHS["B_16"], B=2, T=4, seed=int(f["batch_seed"]), n_trans=1)
    keep = torch.arange(196).unsqueeze(0)
    with torch.no_grad():
        te, ve, pred = O.model_forward(P, {"text": b["text"], "video": b["video"], "keep_ind": keep.expand(2, -1)}, arch)
        assert pred is None
        assert relerr(f["te"], te) < RTOL and relerr(f["ve"], ve) < RTOL
        assert relerr(f["sims"], O.sim_matrix(te, ve)) < RTOL
        cls = O.text_tower(P, torch.tensor(f["prompts"]), arch)
        assert relerr(f["cls_emb"], cls) < RTOL


def test_uint8_transform_tail(golden):
    """CenterCrop -> ClipToTensor -> Normalize of the reference's video_transform.py on uint8 frames: bit-exact."""
    f = golden("transform")
    frames = torch.tensor(f["frames"]).unsqueeze(0)  # [1, T, H0, W0, 3]
    out = O.frames_to_video(frames, int(f["image"]))
    assert torch.equal(out[0], torch.tensor(f["out"]))


def test_downstream_b32(golden):
    """v2/downstream/model_TVTSv2_ViT_B_32.py: 49 unmasked patches per frame."""
    f = golden("downstream_b32")
    arch = dict(O.ARCHS["B_32"], mask_ratio=0.0, sort_head=False)
    P = O.synth_params(arch, seed=int(f["seed"]))
    b = O.synth_batch(O.ARCHS["B_32"], B=2, T=5, seed=int(f["batch_seed"]), n_trans=1)
    with torch.no_grad():
        te, ve, _ = O.model_forward(P, {"text": b["text"], "video": b["video"],
                                        "keep_ind": torch.arange(49).unsqueeze(0).expand(2, -1)}, arch)
    assert relerr(f["te"], te) < RTOL and relerr(f["ve"], ve) < RTOL


def test_model_b16_config2(golden):
    """The headline architecture: the real TVTSv2_B_16 class (tube mask 0.5), B=2, T=4."""
    f = golden("model_b16_cfg2")
    arch = O.ARCHS["B_16"]
    P = leaves(O.synth_params(arch, seed=0))
    b = O.synth_batch(arch, B=2, T=4, seed=int(f["batch_seed"]))
    _check_model(f, arch, P, b)
    sl = {"g_video_proj": ("video_model.proj", (slice(0, 8), slice(0, 16))),
          "g_head": ("pred_model.head.weight", (slice(None), slice(None))),
          "g_cfc11": ("video_model.transformer.resblocks.11.mlp.c_fc.weight", (slice(0, 8), slice(0, 16))),
          "g_tqkv0": ("video_model.transformer.resblocks.0.timeattn.qkv.weight", (slice(0, 8), slice(0, 16))),
          "g_pos": ("video_model.positional_embedding", (slice(None), slice(0, 16))),
          "g_textqkv10": ("text_model.resblocks.10.attn.in_proj_weight", (slice(0, 8), slice(0, 16)))}
    for k, (name, idx) in sl.items():
        assert relerr(f[k], P[name].grad[idx]) < 2e-4, k
    assert relerr(f["g_conv"], P["video_model.conv1.weight"].grad[:4].reshape(4, -1)) < 2e-4


def _alternating_arch():
    return O.tiny_arch(name="small", image=64, patch=16, width=256, heads=4, layers=2, embed=128, text_width=128,
                       text_heads=2, text_layers=12, text_tune_from=9, vocab=512, context=16)


def _alternating_oracle(f, none_grad):
    arch = _alternating_arch()
    P = {k: v.clone() for k, v in O.synth_params(arch, seed=int(f["seed"])).items()}
    state, l1s, l2s = {}, [], []
    for it in range(3):
        yt = O.synth_batch(arch, B=int(f["B"]), T=int(f["T_yt"]), seed=int(f["yt_seeds"][it]), caption_len=int(f["caption_len"]))
        wv = O.synth_batch(arch, B=int(f["B"]), T=int(f["T_wv"]), seed=int(f["wv_seeds"][it]), n_trans=1,
                           caption_len=int(f["caption_len"]))
        for data in (yt, wv):
            l1, l2, _ = O.train_step(P, data, arch, state, none_grad=none_grad)
            l1s.append(l1); l2s.append(l2)
    return P, state, np.array(l1s), np.array(l2s)


def _cut(f, key, t):
    ref = f[key]
    return t[:ref.shape[0], :ref.shape[1]] if (t.dim() == 2 and tuple(t.shape) != ref.shape) else t


def test_alternating_yt_webvid_steps(golden):
    """trainer.py:463-499 over YT (NT = 4) / WebVid (NT = 1) / YT ... : six optimizer steps of the reference's own modules under
    `optimizer.zero_grad()` as the pinned torch 1.11 executes it (zero tensors, not None) and HF AdamW.  In the three WebVid
    steps every `pred_model.*` tensor takes a g = 0 update: moments decay, weights keep moving along m / sqrt(v) and decay."""
    f = golden("alternating_steps")
    P, state, l1s, l2s = _alternating_oracle(f, "zero")
    assert np.allclose(l1s, f["loss1"], atol=2e-5) and np.allclose(l2s, f["loss2"], atol=2e-5), (l1s, f["loss1"], l2s, f["loss2"])
    assert all(s == 6 for s in state["steps"].values())
    for k in f["names"]:
        k = str(k)
        for tag, src in (("p_", P), ("m_", state["m"]), ("v_", state["v"])):
            # parameters are held by their DISPLACEMENT from the initial values (lr 1e-4 / 1e-7 against weights of order 0.1:
            # the values themselves would agree to 1e-5 whatever the optimizer did); measured 3e-7 .. 5e-5, moments 5e-7 .. 2e-6
            got, ref = _cut(f, tag + k, src[k]).detach(), torch.from_numpy(f[tag + k])
            if tag == "p_":
                P0 = _cut(f, tag + k, O.synth_params(_alternating_arch(), seed=int(f["seed"]))[k])
                d_got, d_ref = (got - P0).double(), (ref - P0).double()
                assert float((d_got - d_ref).norm()) <= 5e-4 * float(d_ref.norm()) + 1e-12, (k, float((d_got - d_ref).norm()), float(d_ref.norm()))
            else:
                assert relerr(ref, got) < GTOL, (tag, k, relerr(ref, got))


def test_alternating_steps_fixture_rejects_the_set_to_none_rule(golden):
    """negative control: modern torch's zero_grad(set_to_none=True) -- `pred_model.*` skipped in WebVid steps -- is a different
    trajectory, and the fixture tells them apart by a wide margin (the sort head's first moment is 0.9^-1 too large per skipped
    step, its weights stop moving)."""
    f = golden("alternating_steps")
    P, state, l1s, l2s = _alternating_oracle(f, "skip")
    k = "pred_model.head.weight"
    assert state["steps"][k] == 3
    assert relerr(torch.from_numpy(f["m_" + k]), state["m"][k]) > 0.05
    P0 = O.synth_params(_alternating_arch(), seed=int(f["seed"]))[k]
    d_got, d_ref = (P[k] - P0).double(), (torch.from_numpy(f["p_" + k]) - P0).double()
    assert float((d_got - d_ref).norm()) > 0.2 * float(d_ref.norm())
    assert not np.allclose(l2s, f["loss2"], atol=2e-5)  # the later YT steps see a different sort head
