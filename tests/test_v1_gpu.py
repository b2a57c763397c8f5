"""v1 TVTS step on the HIP engine (run with -m gpu): the kernels it adds (tubelet im2col, per-tube
tube masks in the token assembly, padded-key masking in FULL attention, ReLU) against torch, and the model against the v1
oracle and the fixtures produced by the real v1 classes (tests/golden/v1_*.npz).  Tolerances: SURVEY.md 8d bf16 gates."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tvts_oracle as O  # noqa: E402  (checker only)
from oracle import tvts_v1_oracle as V  # noqa: E402

DEV = "cuda:0"
ARGS = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tvts_amd import hip
    return hip


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def min_cos(a, b):
    a, b = a.detach().double().cpu().reshape(a.shape[0], -1), b.detach().double().cpu().reshape(b.shape[0], -1)
    return float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())


# ------------------------------------------------------------------------------------------------ kernels
def test_tubelet_gather_and_per_tube_assemble(K):
    a = V.tiny_arch()
    P = V.synth_params(a, seed=1)
    batch = V.synth_batch(a, B=3, T=6, seed=2)
    B, T, tb, p, W = 3, 6, a["tubelet"], a["patch"], a["width"]
    tubes, n = T // tb, batch["keep_ind"].shape[2]
    keep = batch["keep_ind"].to(torch.int32).to(DEV)
    cols = torch.full((B * tubes * n, 3 * tb * p * p), float("nan"), dtype=torch.bfloat16, device=DEV)
    K.patch_gather_tube(batch["video"].to(DEV), keep, cols, B=B, tubes=tubes, tubelet=tb, n=n, img=a["image"], patch=p)
    # reference im2col of the kept patches in the Conv3d weight's (c, t, py, px) order
    g = a["image"] // p
    x = batch["video"].reshape(B, tubes, tb, 3, g, p, g, p).permute(0, 1, 4, 6, 3, 2, 5, 7).reshape(B, tubes, g * g, -1)
    ref = torch.gather(x, 2, batch["keep_ind"][..., None].expand(-1, -1, -1, x.shape[-1])).reshape(B * tubes * n, -1)
    assert torch.equal(cols.float().cpu(), ref.bfloat16().float())
    # token assembly with one mask per tube == the oracle's video_tokens (fp32 patch embedding done in torch here)
    pe = (ref @ P["video_model.patch_embed.proj.weight"].reshape(W, -1).t() + P["video_model.patch_embed.proj.bias"]).to(DEV)
    tok = torch.empty(B * (1 + tubes * n), W, device=DEV)
    K.vit_assemble(pe, P["video_model.cls_token"].view(W).to(DEV), P["video_model.pos_embed"].view(-1, W).to(DEV),
                   P["video_model.temporal_embed"].view(-1, W).to(DEV), keep, tok, B=B, T=tubes, n=n)
    want = V.video_tokens(P, batch["video"], batch["keep_ind"], a).reshape(-1, W)
    assert rel(tok, want) < 1e-6
    # backward: scatter of d tok into d patch rows, d pos (per-tube indices), d temporal, d cls
    dtok = torch.randn(tok.shape, generator=torch.Generator().manual_seed(3))
    dpatch = torch.empty(B * tubes * n, W, dtype=torch.bfloat16, device=DEV)
    dcls, dpos, dtemp = torch.zeros(W, device=DEV), torch.zeros(g * g + 1, W, device=DEV), torch.zeros(a["num_frames"] // tb, W, device=DEV)
    K.vit_assemble_bwd(dtok.to(DEV), keep, dpatch, dcls, dpos, dtemp, B=B, T=tubes, n=n)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    V.video_tokens(Pr, batch["video"], batch["keep_ind"], a).reshape(-1, W).backward(dtok)
    assert rel(dpos, Pr["video_model.pos_embed"].grad[0]) < 1e-5 and rel(dtemp, Pr["video_model.temporal_embed"].grad[0]) < 1e-5
    assert rel(dcls, Pr["video_model.cls_token"].grad.view(W)) < 1e-5
    d3 = dtok.view(B, 1 + tubes * n, W)[:, 1:].reshape(B * tubes * n, W)
    assert rel(dpatch.float(), d3) < 4e-3


@pytest.mark.parametrize("B,h,S", [(5, 2, 14), (3, 12, 50), (4, 4, 77), (2, 2, 130)])
def test_full_attention_with_padded_keys(K, B, h, S):
    """FULL attention over right-padded sequences (DistilBERT's attention_mask): forward + backward against torch with the
    masked keys at -inf; padded QUERY rows still produce rows (they are never read), their dK / dV stay exactly zero."""
    dh, W = 64, h * 64
    g = torch.Generator().manual_seed(S)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    qkv = (torch.randn(B * S, 3 * W, generator=g) * 0.5).bfloat16()
    dO = torch.randn(B * S, W, generator=g).bfloat16()
    for b in range(B):  # no gradient arrives at padded query rows in the model
        dO[b * S + int(lens[b]):(b + 1) * S] = 0
    kv = lens.to(torch.int32).to(DEV)
    out = torch.empty(B * S, W, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * S, h, device=DEV)
    K.attn_fwd_len(qkv.to(DEV), kv, out, lse, B=B, heads=h, S=S)
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = (x[:, i * W:(i + 1) * W].reshape(B, S, h, dh).permute(0, 2, 1, 3) for i in range(3))
    s = (q * dh ** -0.5) @ k.transpose(-1, -2)
    s = s.masked_fill((torch.arange(S)[None, :] >= lens[:, None])[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, W)
    valid = (torch.arange(S)[None, :] < lens[:, None]).reshape(-1)
    assert rel(out[valid.to(DEV)].float(), ref[valid]) < 8e-3
    ref.backward(dO.float())
    dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B * S, h, device=DEV)
    K.attn_bwd_len(qkv.to(DEV), kv, dO.to(DEV), out, lse, delta, dqkv, B=B, heads=h, S=S)
    assert torch.isfinite(dqkv.float()).all()
    assert rel(dqkv.float(), x.grad) < 2e-2, rel(dqkv.float(), x.grad)
    assert float(dqkv.float()[~valid.to(DEV)][:, W:].abs().max()) == 0.0 if (~valid).any() else True


def _seed_tensor(v):
    return torch.tensor([v - (1 << 64) if v >= (1 << 63) else v], dtype=torch.int64, device=DEV)


@pytest.mark.parametrize("B,h,S,p", [(5, 2, 14, 0.1), (3, 12, 50, 0.1), (4, 4, 77, 0.25), (2, 2, 130, 0.1)])
def test_attention_probability_dropout(K, B, h, S, p):
    """tvts_attn_fwd_len_drop / bwd_len_drop: weights = dropout(softmax(scores)) with the counter-based mask, forward and backward
    against torch autograd with the SAME mask (oracle drop_mask), padded keys masked, several key tiles (S > 64)."""
    dh, W = 64, h * 64
    g = torch.Generator().manual_seed(S)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    qkv = (torch.randn(B * S, 3 * W, generator=g) * 0.5).bfloat16()
    dO = torch.randn(B * S, W, generator=g).bfloat16()
    for b in range(B):
        dO[b * S + int(lens[b]):(b + 1) * S] = 0
    kv = lens.to(torch.int32).to(DEV)
    seed, site = 0xF00DF00DF00DF00D, 3
    sd = _seed_tensor(seed)
    out = torch.empty(B * S, W, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * S, h, device=DEV)
    K.attn_fwd_len_drop(qkv.to(DEV), kv, out, lse, B=B, heads=h, S=S, p=p, seed=sd, site=site)
    mask = V.drop_mask(seed, site, (B, h, S, S), p)
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = (x[:, i * W:(i + 1) * W].reshape(B, S, h, dh).permute(0, 2, 1, 3) for i in range(3))
    s = (q * dh ** -0.5) @ k.transpose(-1, -2)
    s = s.masked_fill((torch.arange(S)[None, :] >= lens[:, None])[:, None, None, :], float("-inf"))
    ref = ((torch.softmax(s, -1) * mask) @ v).permute(0, 2, 1, 3).reshape(B * S, W)
    valid = (torch.arange(S)[None, :] < lens[:, None]).reshape(-1)
    assert rel(out[valid.to(DEV)].float(), ref[valid]) < 8e-3, rel(out[valid.to(DEV)].float(), ref[valid])
    # the log-sum-exp is of the UNdropped scores (softmax first, dropout second): equal to the plain kernel's
    lse0, out0 = torch.empty_like(lse), torch.empty_like(out)
    K.attn_fwd_len(qkv.to(DEV), kv, out0, lse0, B=B, heads=h, S=S)
    assert float((lse - lse0)[valid.to(DEV)].abs().max()) < 1e-5 and rel(out0[valid.to(DEV)].float(), ref[valid]) > 0.05
    ref.backward(dO.float())
    dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B * S, h, device=DEV)
    K.attn_bwd_len_drop(qkv.to(DEV), kv, dO.to(DEV), out, lse, delta, dqkv, B=B, heads=h, S=S, p=p, seed=sd, site=site)
    assert torch.isfinite(dqkv.float()).all()
    assert rel(dqkv.float(), x.grad) < 2e-2, rel(dqkv.float(), x.grad)
    # p = 0 through the dropout entry points is the plain pair
    out1 = torch.empty_like(out)
    K.attn_fwd_len_drop(qkv.to(DEV), kv, out1, lse, B=B, heads=h, S=S, p=0.0, seed=sd, site=site)
    assert torch.equal(out1[valid.to(DEV)].view(torch.int16), out0[valid.to(DEV)].view(torch.int16))


def test_hidden_state_dropout(K):
    """tvts_dropout_rows: x * m / (1 - p) (+ residual), fp32 and bf16 outputs, bit-exact against the oracle's restatement of the
    generator; keep rate 1 - p; a different site or seed gives a different mask; the seed is read from device memory."""
    M, W, p = 333, 768, 0.1
    g = torch.Generator().manual_seed(0)
    x, res = torch.randn(M, W, generator=g), torch.randn(M, W, generator=g)
    seed = 0x8000000000000123
    sd = _seed_tensor(seed)
    out, outb = torch.empty(M, W, device=DEV), torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    K.dropout_rows(x.to(DEV), p=p, seed=sd, site=4, residual=res.to(DEV), out=out, out_bf16=outb)
    mask = V.drop_mask(seed, 4, (M, W), p)
    want = x * mask + res
    assert torch.equal(out.cpu(), want) and torch.equal(outb.float().cpu(), want.bfloat16().float())
    ones = torch.ones(2000, 512, device=DEV)
    kept = torch.empty_like(ones)
    K.dropout_rows(ones, p=p, seed=sd, site=9, out=kept)
    rate = float((kept > 0).float().mean())
    assert abs(rate - (1 - p)) < 2e-3 and float(kept.max()) == pytest.approx(1 / (1 - p))
    other = torch.empty_like(ones)
    K.dropout_rows(ones, p=p, seed=sd, site=10, out=other)
    assert 0.15 < float((other != kept).float().mean()) < 0.21  # independent masks differ on 2 p (1 - p) = 18 % of the elements
    sd.add_(1)                                                   # a device-side seed update is seen by the next launch
    K.dropout_rows(ones, p=p, seed=sd, site=9, out=other)
    assert 0.15 < float((other != kept).float().mean()) < 0.21


def test_relu(K):
    x = torch.randn(1000, generator=torch.Generator().manual_seed(0)).to(DEV)
    dy = torch.randn(1000, generator=torch.Generator().manual_seed(1)).to(DEV)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    K.relu(x, y)
    K.relu(x, dx, dy=dy)
    assert torch.equal(y, torch.relu(x)) and torch.equal(dx, dy * (x > 0))


# ------------------------------------------------------------------------------------------------ model
def build(a, P, dropout=0.0):
    """dropout 0: the p = 0 model the reference-class fixtures were made with (DistilBertConfig(dropout=0)); the training-mode
    text tower (p = 0.1, the reference's text_model.train()) is covered by the *_dropout tests below"""
    from tvts_amd.model.model_dist_TVTS import TVTS
    m = TVTS(ARGS, arch=dict(a, text_dropout=dropout))
    m.load_state_dict(P, strict=True)
    return m


def engine_step(m, batch):
    from tvts_amd.engine import LossHead
    m._fresh_shadows(); m._sync_requires_grad()
    pb = m.engine.prepare_batch(batch)
    m.store.grad.zero_()
    te, ve, pred = m.engine.forward(pb)
    head = LossHead(m.store.device)
    loss1, dv, dt = head.contrastive(ve, te)
    loss2, dpred = (head.sorting(pred, batch["label"].reshape(-1).to(torch.int32).to(DEV)) if pred is not None else (None, None))
    m.engine.backward(dt, dv, dpred)
    torch.cuda.synchronize()
    return float(loss1), (float(loss2) if loss2 is not None else 0.0), te.clone(), ve.clone(), pred


def check_grads(store, grads, gn_tol=0.01, cos_tol=0.985):
    # per-tensor cosine gate: v2's tests hold 0.995; the tiny v1 model's worst tensor is a 128-element DistilBERT bias (v_lin.bias of
    # layer 0, NT = 1) at 0.990 -- a handful of bf16 roundings against 128 numbers -- every weight matrix measures >= 0.998
    tot_ref = sum(float(g.norm()) ** 2 for g in grads.values()) ** 0.5
    tot, worst = 0.0, []
    for k, g in grads.items():
        mine = store.g(k).detach().cpu()
        assert torch.isfinite(mine).all(), k
        tot += float(mine.norm()) ** 2
        if float(g.norm()) > 1e-3 * tot_ref:
            worst.append((float(torch.nn.functional.cosine_similarity(mine.double().flatten(), g.double().flatten(), dim=0)), k,
                          float(mine.norm()), float(g.norm())))
    worst.sort()
    assert abs(tot ** 0.5 - tot_ref) < gn_tol * tot_ref, (tot ** 0.5, tot_ref, worst[:5])
    assert worst[0][0] > cos_tol, worst[:8]


def small():
    from tvts_amd import arch as A
    a = A.small_arch_v1()
    return a, V.tiny_arch(**{k: a[k] for k in V.tiny_arch() if k in a and k != "name"})


@pytest.mark.parametrize("nt", [4, 1])
def test_small_v1_forward_backward(K, nt):
    a, oa = small()
    P = V.synth_params(oa, seed=5)
    batch = V.synth_batch(oa, B=4, T=6, seed=6, n_trans=nt, caption_len=13)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    r1, r2, rte, rve, rpred = V.step_losses(leaves, batch, oa)
    (r1 + r2).backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    m = build(a, P)
    assert list(dict(m.named_parameters()).keys()) == list(V.param_shapes(oa).keys())
    l1, l2, te, ve, pred = engine_step(m, batch)
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    if nt == 1:
        assert pred is None and rpred is None
        assert float(m.store.g("pred_model.head.weight").abs().max()) == 0.0
    else:
        assert rel(pred.view_as(rpred), rpred) < 0.03
    # one caption per video: no mean over 4 captions, the text embedding carries twice the bf16 noise into a 4 x 4 similarity at
    # temperature 0.05 (embed dim 64 here) -> 2e-2 on the contrastive loss; the embeddings themselves are held to the 2 % gate above
    assert abs(l1 - float(r1)) < (1e-2 if nt == 4 else 2e-2) and abs(l2 - float(r2)) < 1e-2, (l1, float(r1), l2, float(r2))
    check_grads(m.store, grads, gn_tol=0.01 if nt == 4 else 0.03)
    # the pieces the reference exposes (model_dist_TVTS.py:131-147)
    tb, t = m.compute_text(batch["text"])
    rb, rt = V.compute_text(P, batch["text"], oa)
    assert rel(tb, rb) < 0.02 and rel(t, rt) < 0.02
    vb, v = m.compute_video(batch["video"], batch["keep_ind"])
    rvb, rv = V.compute_video(P, batch["video"], batch["keep_ind"], oa)
    assert vb.shape == rvb.shape and rel(vb, rvb) < 0.02 and rel(v, rv) < 0.02


@pytest.mark.parametrize("nt", [4, 1])
def test_small_v1_training_mode_dropout(K, nt):
    """The training step the reference runs: DistilBERT in train() mode (v1/model/model_dist_TVTS.py:33-34), dropout 0.1 on the
    embedding output, the attention probabilities and the FFN output.  The engine draws counter-based masks (regenerated in the
    backward); the oracle -- pinned to the real transformers class in this mode by tests/golden/v1_tiny_dropout.npz -- takes the
    same seed: embeddings, losses and every gradient at the bf16 gates of the p = 0 test.  Then: a second step draws NEW masks,
    eval mode draws none."""
    a, oa = small()
    P = V.synth_params(oa, seed=5)
    batch = V.synth_batch(oa, B=4, T=6, seed=6, n_trans=nt, caption_len=13)
    m = build(a, P, dropout=0.1)
    eng = m.engine
    assert eng.training and eng.text_drop_p == 0.1
    seed0 = int(eng.drop_seed.item()) & ((1 << 64) - 1)
    seed1 = (seed0 + eng.DROP_STEP_STRIDE) & ((1 << 64) - 1)   # the engine advances the seed at the start of a training forward
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    r1, r2, rte, rve, rpred = V.step_losses(leaves, batch, oa, drop=dict(p=0.1, seed=seed1))
    (r1 + r2).backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    l1, l2, te, ve, pred = engine_step(m, batch)
    assert (int(eng.drop_seed.item()) & ((1 << 64) - 1)) == seed1
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02
    assert abs(l1 - float(r1)) < (1e-2 if nt == 4 else 2e-2) and abs(l2 - float(r2)) < 1e-2, (l1, float(r1), l2, float(r2))
    check_grads(m.store, grads, gn_tol=0.01 if nt == 4 else 0.03)
    # not the p = 0 model
    with torch.no_grad():
        te0 = V.model_forward(P, batch, oa)[0]
    assert rel(te, te0) > 0.02
    # a second training forward: new masks
    te_a = te.clone()
    _, _, te_b, _, _ = engine_step(m, batch)
    assert rel(te_b, te_a) > 0.02
    # eval mode (validation, retrieval features): no dropout, the module flag is mirrored into the engine
    m.eval()
    with torch.no_grad():
        te_e, ve_e, _ = m(batch)
    assert not eng.training and rel(te_e, te0) < 0.02 and min_cos(te_e, te0) > 0.9995
    m.train()
    te_t, _, _ = m(batch)
    assert eng.training and rel(te_t.detach(), te0) > 0.02


def test_v1_training_curve_with_dropout_tracks_oracle(K):
    """ten optimizer steps with the training-mode text tower: the engine's seed sequence handed to the oracle step by step"""
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    a, oa = small()
    P = V.synth_params(oa, seed=7)
    batch = V.synth_batch(oa, B=4, T=4, seed=8, caption_len=9)
    m = build(a, P, dropout=0.1)
    opt = FusedHFAdamW([dict(params=list(m.parameters()), lr=3e-4, weight_decay=0.0)], m.store, model=m)
    run = StepRunner(m, opt)
    seed = int(m.engine.drop_seed.item()) & ((1 << 64) - 1)
    Pr = {k: v.clone() for k, v in P.items()}
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in Pr.items()}
    curve, ref = [], []
    for step in range(1, 11):
        seed = (seed + m.engine.DROP_STEP_STRIDE) & ((1 << 64) - 1)
        leaves = {k: v.clone().requires_grad_(True) for k, v in Pr.items()}
        r1, r2, *_ = V.step_losses(leaves, batch, oa, drop=dict(p=0.1, seed=seed))
        (r1 + r2).backward()
        ref.append(float(r1 + r2))
        for k in Pr:
            gk = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(Pr[k])
            O.hf_adamw_step(Pr[k], gk, st[k][0], st[k][1], step, 3e-4, 0.0)
        out = run.step(batch)
        curve.append(float(out["loss1"]) + float(out["loss2"]))
    curve, ref = np.array(curve), np.array(ref)
    assert ref[-1] < ref[0] - 0.02, ref
    assert np.all(np.abs(curve - ref) < 0.02 * np.abs(ref) + 1e-2), (curve, ref)


def test_v1_full_size_against_reference_golden(K, golden):
    """the real v1 TVTS class (DistilBERT-base + tubelet ViT-B/16 + sorting head, 170 M parameters) ran in the build container
    at B=2, 4 frames, mask 0.75; same synthetic parameters and batch here"""
    f = golden("v1_full")
    a = V.ARCH
    from tvts_amd import arch as A
    P = V.synth_params(a, seed=int(f["seed"]))
    m = build(A.ARCH_V1, P)
    del P
    batch = V.synth_batch(a, B=int(f["B"]), T=int(f["T"]), seed=int(f["batch_seed"]), caption_len=int(f["caption_len"]))
    l1, l2, te, ve, pred = engine_step(m, batch)
    rte, rve, rpred = torch.tensor(f["te"]), torch.tensor(f["ve"]), torch.tensor(f["pred"])
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    assert float((pred.view_as(rpred).cpu() - rpred).abs().max()) < 0.05
    assert abs(l1 - float(f["loss1"])) < 1e-2 and abs(l2 - float(f["loss2"])) < 1e-2, (l1, l2)
    gn = float(m.store.grad.double().norm())
    assert abs(gn - float(f["grad_norm"])) < 0.01 * float(f["grad_norm"]), (gn, float(f["grad_norm"]))
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    bad = []
    for k, v in ref.items():
        mine = float(m.store.g(k).double().norm())
        if float(v) > 1e-3 * float(f["grad_norm"]) and abs(mine - float(v)) > 0.05 * float(v):
            bad.append((k, mine, float(v)))
    assert not bad, bad[:10]
    for key, (name, idx) in {"g_conv": ("video_model.patch_embed.proj.weight", (slice(0, 2),)),
                             "g_qkv11": ("video_model.blocks.11.attn.qkv.weight", (slice(0, 8), slice(0, 32))),
                             "g_qlin5": ("text_model.transformer.layer.5.attention.q_lin.weight", (slice(0, 8), slice(0, 32))),
                             "g_txtproj": ("txt_proj.1.weight", (slice(0, 8), slice(0, 32))),
                             "g_vidproj": ("vid_proj.0.weight", (slice(0, 8), slice(0, 32))),
                             "g_head": ("pred_model.head.weight", (slice(None), slice(0, 64)))}.items():
        assert rel(m.store.g(name)[idx], torch.tensor(f[key])) < 0.1, (key, rel(m.store.g(name)[idx], torch.tensor(f[key])))


def test_v1_training_curve_tracks_oracle(K):
    """10 optimizer steps on a fixed batch: the fused HF-AdamW with the v1 entrypoint's single parameter group (lr 1e-4 x 3,
    weight_decay 0, v1/configs/dist-yt-pt.json) against the oracle's restated HF AdamW."""
    from tvts_amd import arch as A
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    a, oa = small()
    P = V.synth_params(oa, seed=7)
    batch = V.synth_batch(oa, B=4, T=4, seed=8, caption_len=9)
    m = build(a, P)
    assert all(A.param_group_of(n, a) == 0 for n in m.store.shapes)
    opt = FusedHFAdamW([dict(params=list(m.parameters()), lr=3e-4, weight_decay=0.0)], m.store, model=m)
    run = StepRunner(m, opt)
    Pr = {k: v.clone() for k, v in P.items()}
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in Pr.items()}
    curve, ref = [], []
    for step in range(1, 11):
        leaves = {k: v.clone().requires_grad_(True) for k, v in Pr.items()}
        r1, r2, *_ = V.step_losses(leaves, batch, oa)
        (r1 + r2).backward()
        ref.append(float(r1 + r2))
        for k in Pr:
            gk = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(Pr[k])
            O.hf_adamw_step(Pr[k], gk, st[k][0], st[k][1], step, 3e-4, 0.0)
        out = run.step(batch)
        curve.append(float(out["loss1"]) + float(out["loss2"]))
    curve, ref = np.array(curve), np.array(ref)
    assert ref[-1] < ref[0] - 0.02, ref
    assert np.all(np.abs(curve - ref) < 0.02 * np.abs(ref) + 1e-2), (curve, ref)


class _V1Loader(list):
    def __init__(self, batches):
        super().__init__(batches)
        self.dataset_name, self.batch_size, self.n_samples = "YTTemporal", 4, 4 * len(batches)


class _V1Config(dict):  # (module level: the checkpoint pickles the config object)
    resume = None

    def __init__(self, save_dir, epochs):
        super().__init__(trainer=dict(epochs=epochs, save_period=1, verbosity=2, monitor="off", init_val=False),
                         arch=dict(type="TVTS", args={}), optimizer=dict(type="AdamW", args=dict(lr=1e-4)))
        self.save_dir = save_dir

    def get_logger(self, name, verbosity=2):
        import logging
        return logging.getLogger(name)


def test_v1_resumed_run_draws_the_masks_of_the_uninterrupted_run(K, tmp_path):
    """Trainer_TVTS keeps the dropout generator's seed in the checkpoint (beside the reference's six keys): two epochs in one go and
    one epoch + checkpoint + resume + one epoch end in the same parameters bit for bit, with the training-mode DistilBERT dropout on.
    The seed of a data-parallel rank is mixed with the rank, so that ranks draw different masks like the reference's do."""
    from tvts_amd.model.loss import NormSoftmaxLoss
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.trainer.trainer import Trainer_TVTS
    a, oa = small()

    def make(save_dir, epochs, resume=None):
        m = build(a, V.synth_params(oa, seed=7), dropout=0.1)
        opt = FusedHFAdamW([dict(params=list(m.parameters()), lr=3e-4, weight_decay=0.0)], m.store, model=m)
        cfg = _V1Config(str(save_dir), epochs)
        cfg.resume = resume
        dl = _V1Loader([V.synth_batch(oa, B=4, T=4, seed=40 + i, caption_len=9) for i in range(3)])
        args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1, schedule=[])
        return Trainer_TVTS(args, m, NormSoftmaxLoss(), [], opt, config=cfg, data_loader=[dl], valid_data_loader=None), m

    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    tr_a, m_a = make(tmp_path / "a", 2)
    seed0 = int(m_a.engine.drop_seed.item())
    tr_a.train()
    assert int(m_a.engine.drop_seed.item()) != seed0
    tr_b, m_b = make(tmp_path / "b", 1)
    tr_b.train()
    ck = torch.load(tmp_path / "b" / "checkpoint-epoch1.pth", map_location="cpu", weights_only=False)
    assert ck["tvts_amd"]["drop_seed_base"] == int(m_b.engine.drop_seed.item()) & ((1 << 64) - 1)  # rank 0: offset 0
    assert list(ck.keys())[:6] == ["arch", "epoch", "state_dict", "optimizer", "monitor_best", "config"]
    tr_c, m_c = make(tmp_path / "b", 2, resume=tmp_path / "b" / "checkpoint-epoch1.pth")
    assert int(m_c.engine.drop_seed.item()) == int(m_b.engine.drop_seed.item())
    tr_c.train()
    assert torch.equal(m_c.store.flat, m_a.store.flat) and torch.equal(m_c.store.m, m_a.store.m)
    # rank mixing: the same arch on "rank 3" starts from another seed
    import tvts_amd.dist as D
    orig = D.world
    try:
        D.world = lambda: (8, 3)
        m_r = build(a, V.synth_params(oa, seed=7), dropout=0.1)
        assert int(m_r.engine.drop_seed.item()) != seed0
        # a resume on "rank 3" of the file rank 0 wrote continues rank 3's OWN sequence: base + 3 strides, not rank 0's seed
        base = ck["tvts_amd"]["drop_seed_base"]
        tr_r, m_r2 = make(tmp_path / "b", 2, resume=tmp_path / "b" / "checkpoint-epoch1.pth")
        got = int(m_r2.engine.drop_seed.item()) & ((1 << 64) - 1)
        assert got == (base + 3 * m_r2.engine.DROP_RANK_STRIDE) & ((1 << 64) - 1) != base
        assert m_r2.engine.drop_seed_base() == base
    finally:
        D.world = orig
