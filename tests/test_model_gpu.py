"""End-to-end GPU parity of the HIP step engine against the fp32 CPU oracle and the committed goldens
(run with -m gpu).  Tolerances are the bf16 gates of SURVEY.md 8d: per-row cosine >= 0.9995 and rel-L2 <= 2 %
on embeddings, |d loss| <= 1e-2, grad-norm within 1 %, 20-step loss curves within 2 %."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tvts_oracle as O  # noqa: E402  (checker only)

DEV = "cuda:0"
ARGS = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return True


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def min_cos(a, b):
    a, b = a.detach().double().cpu().reshape(a.shape[0], -1), b.detach().double().cpu().reshape(b.shape[0], -1)
    return float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())


def build(arch_name=None, arch=None, seed=0):
    from tvts_amd import arch as A
    from tvts_amd.model._common import TVTSv2Base
    a = dict(arch) if arch is not None else A.ARCHS[arch_name]
    oarch = O.tiny_arch(**a) if arch is not None else O.ARCHS[arch_name]
    P = O.synth_params(oarch, seed=seed)
    m = TVTSv2Base(ARGS, arch=a)
    m.load_state_dict(P, strict=True)
    return m, oarch, P


def oracle_step(P, batch, oarch):
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    l1, l2, te, ve, pred = O.step_losses(leaves, batch, oarch)
    (l1 + l2).backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return float(l1), float(l2), te.detach(), ve.detach(), None if pred is None else pred.detach(), grads


def engine_step(m, batch):
    """forward + fused losses + hand-written backward; returns everything, no optimizer step."""
    from tvts_amd.engine import LossHead
    m._fresh_shadows(); m._sync_requires_grad()
    eng = m.engine
    pb = eng.prepare_batch(batch)
    m.store.grad.zero_()
    te, ve, pred = eng.forward(pb)
    head = LossHead(m.store.device)
    loss1, dv, dt = head.contrastive(ve, te)
    loss2, dpred = (head.sorting(pred, batch["label"].reshape(-1).to(torch.int32).to(DEV)) if pred is not None else (None, None))
    eng.backward(dt, dv, dpred)
    torch.cuda.synchronize()
    return float(loss1), (float(loss2) if loss2 is not None else 0.0), te.clone(), ve.clone(), pred, m.store


def check_grads(store, grads, gn_tol=0.01, cos_tol=0.995):
    tot_ref = sum(float(g.norm()) ** 2 for g in grads.values()) ** 0.5
    tot = 0.0
    worst = []
    for k, g in grads.items():
        mine = store.g(k).detach().cpu()
        assert torch.isfinite(mine).all(), k
        tot += float(mine.norm()) ** 2
        if float(g.norm()) > 1e-3 * tot_ref:
            c = float(torch.nn.functional.cosine_similarity(mine.double().flatten(), g.double().flatten(), dim=0))
            worst.append((c, k, float(mine.norm()), float(g.norm())))
    tot = tot ** 0.5
    worst.sort()
    assert abs(tot - tot_ref) < gn_tol * tot_ref, (tot, tot_ref, worst[:5])
    assert worst[0][0] > cos_tol, worst[:8]
    if gn_tol > 0.01:  # the e4m3 paths: report the margins their (wider) gates leave (pytest -s)
        print(f"   [margins] gradient norm {tot / tot_ref - 1:+.3%} (gate {gn_tol:.0%}), worst tensor cosines "
              + ", ".join(f"{c:.4f} {k.split('resblocks.')[-1]}" for c, k, *_ in worst[:3]) + f" (gate {cos_tol})")
    return tot, tot_ref, worst


def test_small_arch_forward_backward(gpu):
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch(), seed=3)
    batch = O.synth_batch(oarch, B=4, T=3, seed=5, caption_len=11)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02
    assert rel(pred.view_as(rpred), rpred) < 0.03
    assert abs(l1 - r1) < 1e-2 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
    check_grads(store, grads)


def test_captions_of_different_lengths(gpu):
    """What clip.tokenize hands the step (trainer.py:465-475): captions of different lengths in one batch, zero-padded behind their
    end token.  The engine trims the context to the longest caption of the batch and reads every caption at ITS end token
    (argmax of the ids, CLIP/clip/model.py:354); the padded rows behind it can, by causality, neither reach the embedding nor
    receive a gradient (token 0's embedding row and the positions behind the shortest caption get exactly what the oracle gives)."""
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch(), seed=13)
    batch = O.synth_batch(oarch, B=6, T=3, seed=15, caption_len=12)
    eot = oarch["vocab"] - 1
    g = torch.Generator().manual_seed(16)
    lens = torch.randint(3, 13, (batch["text"].shape[0],), generator=g)
    lens[0], lens[1] = 12, 3  # the longest and the shortest possible
    for r, n in enumerate(lens.tolist()):
        batch["text"][r, n - 1] = eot
        batch["text"][r, n:] = 0
    assert (batch["text"].argmax(-1) == lens - 1).all()
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02
    assert abs(l1 - r1) < 1e-2 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
    check_grads(store, grads)
    demb, rdemb = store.g("text_token_embedding.weight").cpu(), grads["text_token_embedding.weight"]
    assert float(rdemb[0].abs().max()) == 0.0 and float(demb[0].abs().max()) == 0.0  # the padding token learns nothing
    dpos, rdpos = store.g("text_positional_embedding").cpu(), grads["text_positional_embedding"]
    assert float(dpos[12:].abs().max()) == 0.0 and rel(dpos[:12], rdpos[:12]) < 0.05


def test_full_context_captions_and_a_batch_of_one(gpu):
    """The two ends of the input contract on the real ViT-B/16: captions that fill CLIP's whole context (77 tokens, end token in the
    last column: the text tower's attention runs on 77-token sequences instead of the 32-token fused path) mixed with short ones,
    and a batch of ONE pair (the similarity matrix is 1 x 1, the contrastive loss zero, the sorting loss still trains)."""
    m, oarch, P = build("B_16", seed=23)
    batch = O.synth_batch(oarch, B=2, T=2, seed=24, caption_len=77)
    assert int(batch["text"].argmax(-1).min()) == 76
    eot = oarch["vocab"] - 1
    for r, n in ((1, 5), (4, 33), (6, 40)):
        batch["text"][r, n - 1] = eot
        batch["text"][r, n:] = 0
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02
    assert abs(l1 - r1) < 1e-2 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
    check_grads(store, grads)
    one = {k: (v[:1] if k != "text" else v.reshape(oarch["n_trans"], 2, -1)[:, :1].reshape(oarch["n_trans"], -1)) for k, v in batch.items()}
    r1, r2, rte, rve, rpred, grads = oracle_step(P, one, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, one)
    assert abs(r1) < 1e-6 and abs(l1) < 1e-6 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
    assert min_cos(te, rte) > 0.9995 and min_cos(ve, rve) > 0.9995
    check_grads(store, grads)


@pytest.mark.parametrize("T", [1, "max"])
def test_one_frame_and_the_whole_temporal_table(gpu, T):
    """clips of ONE frame (the time attention sees the frame's own token and the class token only) and of as many frames as the
    temporal embedding has rows (num_frames, video_encoder_ViT_B_16.py:190-203)"""
    from tvts_amd import arch as A
    a = A.small_arch()
    T = a["num_frames"] if T == "max" else T
    m, oarch, P = build(arch=a, seed=31)
    batch = O.synth_batch(oarch, B=3, T=T, seed=32, caption_len=10)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    assert rel(pred.view_as(rpred), rpred) < 0.03
    assert abs(l1 - r1) < 1e-2 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
    check_grads(store, grads)
    dt, rdt = store.g("video_model.temporal_embedding").cpu(), grads["video_model.temporal_embedding"]
    assert float(dt[T:].abs().max() if T < dt.shape[0] else 0.0) == 0.0 and rel(dt[:T], rdt[:T]) < 0.05


def test_malformed_batches_fail_loudly(gpu):
    """what the reference answers with an indexing / broadcasting error must not become a read or write past a table"""
    from tvts_amd import arch as A
    a = A.small_arch()
    m, oarch, P = build(arch=a, seed=41)
    good = O.synth_batch(oarch, B=2, T=2, seed=42, caption_len=8)
    m.engine.prepare_batch(good)

    def bad(**kw):
        b = {k: v.clone() for k, v in good.items()}
        b.update(kw)
        return b
    ppf = (a["image"] // a["patch"]) ** 2
    k = good["keep_ind"].clone(); k[0, 0] = ppf
    with pytest.raises(IndexError):
        m.engine.prepare_batch(bad(keep_ind=k))
    with pytest.raises(ValueError):
        m.engine.prepare_batch(bad(keep_ind=good["keep_ind"][:1].repeat(3, 1)))
    t = good["text"].clone(); t[0, 1] = a["vocab"]
    with pytest.raises(IndexError):
        m.engine.prepare_batch(bad(text=t))
    with pytest.raises(ValueError):
        m.engine.prepare_batch(bad(text=good["text"][:-1]))
    long_clip = O.synth_batch(oarch, B=2, T=a["num_frames"] + 1, seed=43, caption_len=8)
    with pytest.raises(ValueError):
        m.engine.prepare_batch(long_clip)


def test_small_arch_webvid_batch(gpu):
    """NT = 1: no sorting head, pred None, pred_model receives no gradient (trainer.py:494)."""
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch(), seed=4)
    batch = O.synth_batch(oarch, B=4, T=3, seed=6, n_trans=1, caption_len=9)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert pred is None and rpred is None
    assert min_cos(ve, rve) > 0.9995 and abs(l1 - r1) < 1e-2
    check_grads(store, grads)
    assert float(store.g("pred_model.head.weight").abs().max()) == 0.0


def test_small_h14_forward_backward(gpu):
    """H/14 structure: head dim 80, 14x14 patches (K 588 -> 640), erf-GELU, pooled = ln_post(CLS) @ proj, the sort head
    sees the un-normalised patch tokens without CLS (video_encoder_ViT_H_14.py:472-484)."""
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch_h(), seed=5)
    batch = O.synth_batch(oarch, B=4, T=3, seed=7, caption_len=11)
    assert batch["keep_ind"].shape[1] == 4
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    assert rel(pred.view_as(rpred), rpred) < 0.03
    assert abs(l1 - r1) < 1e-2 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
    check_grads(store, grads)
    assert list(dict(m.named_parameters()).keys()) == list(O.param_shapes(oarch).keys())
    # the pieces the reference exposes: (patch tokens @ proj without CLS, pooled embedding)
    tok, pooled = m.compute_video(batch["video"], batch["keep_ind"])
    assert tok.shape == (4, 3 * 4, oarch["embed"]) and rel(pooled, rve) < 0.02


def test_small_h14_webvid_batch_16_frames(gpu):
    """NT = 1 (no sorting loss: the patch-token projection gets no gradient) and a 16-frame clip, which needs the
    temporal table widened past the reference's 12 rows (BASELINE config 3)."""
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch_h(num_frames=16), seed=6)
    batch = O.synth_batch(oarch, B=3, T=16, seed=8, n_trans=1, caption_len=9)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert pred is None and rpred is None
    assert min_cos(ve, rve) > 0.9995 and abs(l1 - r1) < 1e-2
    # B = 3 at temperature 0.05: the softmax gradient scales every parameter gradient by the same factor, which moves
    # ~1.4 % with the bf16 forward perturbation here (per-tensor cosines stay > 0.9997) -> 2 % on the norm
    check_grads(store, grads, gn_tol=0.02)


def test_autograd_surface_matches_engine(gpu):
    """The nn.Module path the kept entrypoint / reference trainer would drive: model(data), sim_matrix,
    NormSoftmaxLoss, CE*2, loss.backward(), p.grad."""
    from tvts_amd import arch as A
    from tvts_amd.model._common import sim_matrix
    from tvts_amd.model.loss import NormSoftmaxLoss
    m, oarch, P = build(arch=A.small_arch(), seed=3)
    batch = O.synth_batch(oarch, B=4, T=3, seed=5, caption_len=11)
    l1e, l2e, *_ = engine_step(m, batch)
    ref = {k: m.store.g(k).clone() for k in ("video_model.proj", "text_projection", "pred_model.head.weight",
                                             "video_model.transformer.resblocks.0.timeattn.qkv.weight")}
    m.store.grad.zero_()
    te, ve, pred = m(batch)
    loss1 = NormSoftmaxLoss()(sim_matrix(ve, te))
    loss2 = torch.nn.CrossEntropyLoss()(pred.reshape(-1, 4), batch["label"].reshape(-1).to(DEV)) * 2
    (loss1 + loss2).backward()
    assert abs(float(loss1) - l1e) < 1e-4 and abs(float(loss2) - l2e) < 1e-4
    pd = dict(m.named_parameters())
    for k, g in ref.items():
        assert pd[k].grad is not None and rel(pd[k].grad, g) < 1.5e-2, k  # bf16 re-rounding noise between two runs
    assert list(pd.keys()) == list(O.param_shapes(oarch).keys())


def test_autograd_path_with_a_stock_optimizer_does_not_accumulate_stale_gradients(gpu):
    """model(data) -> loss.backward() -> torch.optim.AdamW.step() -> zero_grad() (set_to_none=True by default): the flat
    gradient buffer must restart from zero on the next backward, and keep accumulating when .grad is left in place."""
    from tvts_amd import arch as A
    from tvts_amd.model._common import sim_matrix
    from tvts_amd.model.loss import NormSoftmaxLoss
    m, oarch, P = build(arch=A.small_arch(), seed=3)
    batch = O.synth_batch(oarch, B=4, T=3, seed=5, caption_len=11)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=0.0)  # lr 0: both steps see the same weights
    crit = NormSoftmaxLoss()

    def fwd_bwd():
        te, ve, pred = m(batch)
        loss = crit(sim_matrix(ve, te)) + 2 * torch.nn.CrossEntropyLoss()(pred.reshape(-1, 4), batch["label"].reshape(-1).to(DEV))
        loss.backward()
    fwd_bwd()
    pd = dict(m.named_parameters())
    keys = ("video_model.proj", "text_projection", "pred_model.head.weight")
    g1 = {k: pd[k].grad.clone() for k in keys}
    opt.step()
    opt.zero_grad()  # default set_to_none=True
    assert pd["video_model.proj"].grad is None
    fwd_bwd()
    for k in keys:
        assert rel(pd[k].grad, g1[k]) < 1e-5, k  # NOT 2 x g1
    fwd_bwd()  # no zero_grad in between: autograd accumulation
    for k in keys:
        assert rel(pd[k].grad, 2 * g1[k]) < 1e-5, k
    # a foreign .grad tensor (not a view of the flat buffer) receives this backward's gradient by addition
    opt.zero_grad()
    pd["text_projection"].grad = torch.ones_like(pd["text_projection"])
    fwd_bwd()
    assert rel(pd["text_projection"].grad, 1 + g1["text_projection"]) < 1e-5
    assert rel(pd["video_model.proj"].grad, g1["video_model.proj"]) < 1e-5


def test_frozen_text_layers(gpu):
    """requires_grad=False on text resblocks below the tune range (train_dist..:89-96): no wgrad, dgrad still flows."""
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch(), seed=3)
    for n_, p in m.named_parameters():
        if n_.startswith("text_model.resblocks.0."):
            p.requires_grad = False
    batch = O.synth_batch(oarch, B=4, T=3, seed=5, caption_len=11)
    _, _, _, _, _, grads = oracle_step(P, batch, oarch)
    engine_step(m, batch)
    assert float(m.store.g("text_model.resblocks.0.mlp.c_fc.weight").abs().max()) == 0.0
    k = "text_token_embedding.weight"
    assert rel(m.store.g(k), grads[k]) < 0.05


def test_training_curve_tracks_oracle(gpu):
    """20 optimizer steps on a fixed batch, fused HF-AdamW vs the oracle's restated HF AdamW (lr x3 so that
    20 steps move the loss)."""
    from tvts_amd import arch as A
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    a = A.small_arch()
    m, oarch, P = build(arch=a, seed=7)
    batch = O.synth_batch(oarch, B=4, T=2, seed=8, caption_len=9)
    hp = [(lr * 3, wd) for lr, wd in A.GROUP_HPARAMS]
    groups = [[], [], [], []]
    for name, p in m.named_parameters():
        gi = A.param_group_of(name, a)
        if gi < 0:
            p.requires_grad = False
        else:
            groups[gi].append(p)
    opt = FusedHFAdamW([dict(params=groups[i], lr=hp[i][0], weight_decay=hp[i][1]) for i in range(4)], m.store, model=m)
    runner = StepRunner(m, opt)
    Pr = {k: v.clone() for k, v in P.items()}
    state, O_HP = {}, O.GROUP_HPARAMS
    O.GROUP_HPARAMS = tuple(hp)
    try:
        ref_curve, curve = [], []
        for i in range(20):
            r1, r2, _ = O.train_step(Pr, batch, oarch, state)
            ref_curve.append(r1 + r2)
            out = runner.step(batch)
            curve.append(float(out["loss1"]) + float(out["loss2"]))
    finally:
        O.GROUP_HPARAMS = O_HP
    ref_curve, curve = np.array(ref_curve), np.array(curve)
    assert ref_curve[-1] < ref_curve[0] - 0.02, ref_curve  # the problem actually trains
    assert np.all(np.abs(curve - ref_curve) < 0.02 * np.abs(ref_curve) + 1e-2), (curve, ref_curve)
    # parameters after 20 steps
    for k in ("pred_model.head.weight", "video_model.transformer.resblocks.1.timeattn.proj.weight"):
        assert rel(m.store.p(k), Pr[k]) < 2e-2, k


def test_training_curve_fp8_dgrad_tracks_oracle(gpu):
    """The same 20-step curve on BASELINE config 4's path (H/14 structure, e4m3 forward AND input-gradient GEMMs, multi-tensor weight
    re-quantisation every step) against the oracle that emulates the e4m3 forward: the losses track within 3 %, the problem trains,
    and the curve stays within 2 % of the engine's own bf16-backward curve."""
    from tvts_amd import arch as A
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    hp = [(lr * 3, wd) for lr, wd in A.GROUP_HPARAMS]

    def engine_curve(a, batch):
        m, oarch, P = build(arch=a, seed=7)
        groups = [[], [], [], []]
        for name, p in m.named_parameters():
            gi = A.param_group_of(name, a)
            if gi < 0:
                p.requires_grad = False
            else:
                groups[gi].append(p)
        opt = FusedHFAdamW([dict(params=groups[i], lr=hp[i][0], weight_decay=hp[i][1]) for i in range(4) if groups[i]], m.store, model=m)
        runner = StepRunner(m, opt)
        curve = []
        for i in range(20):
            out = runner.step(batch)
            curve.append(float(out["loss1"]) + float(out["loss2"]))
        return np.array(curve), oarch, P

    a8 = A.small_arch_h(width=640, heads=8, fp8=True, fp8_dgrad=True)
    _, oarch, P = build(arch=a8, seed=7)
    batch = O.synth_batch(oarch, B=4, T=2, seed=8, caption_len=9)
    curve, oarch, P = engine_curve(a8, batch)
    curve_fwd, _, _ = engine_curve(A.small_arch_h(width=640, heads=8, fp8=True), batch)
    # ... and BASELINE config 5 in full: e4m3 weight gradients under per-tensor delayed scales (step 0 calibrates per token)
    curve_w, _, _ = engine_curve(A.small_arch_h(width=640, heads=8, fp8_wgrad=True), batch)
    Pr = {k: v.clone() for k, v in P.items()}
    state, O_HP = {}, O.GROUP_HPARAMS
    O.GROUP_HPARAMS = tuple(hp)
    try:
        ref_curve = []
        for i in range(20):
            r1, r2, _ = O.train_step(Pr, batch, oarch, state)
            ref_curve.append(r1 + r2)
    finally:
        O.GROUP_HPARAMS = O_HP
    ref_curve = np.array(ref_curve)
    assert ref_curve[-1] < ref_curve[0] - 0.02 and curve[-1] < curve[0] - 0.02, (ref_curve, curve)
    assert np.all(np.abs(curve - ref_curve) < 0.03 * np.abs(ref_curve) + 1e-2), (curve, ref_curve)
    assert np.all(np.abs(curve - curve_fwd) < 0.02 * np.abs(curve_fwd) + 1e-2), (curve, curve_fwd)
    assert curve_w[0] == curve[0] and curve_w[-1] < curve_w[0] - 0.02                       # the calibration step is the per-token step
    assert np.all(np.abs(curve_w - ref_curve) < 0.03 * np.abs(ref_curve) + 1e-2), (curve_w, ref_curve)
    assert np.all(np.abs(curve_w - curve) < 0.02 * np.abs(curve) + 1e-2), (curve_w, curve)


def test_b32_config1_against_reference_golden(gpu, golden):
    """BASELINE config 1 (the real TVTSv2_B_32 class ran in the build container): B/32, B=2, T=4."""
    f = golden("model_b32_cfg1")
    m, oarch, P = build(arch_name="B_32", seed=0)
    del P
    batch = O.synth_batch(oarch, B=2, T=4, seed=0)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    rte, rve, rpred = torch.tensor(f["te"]), torch.tensor(f["ve"]), torch.tensor(f["pred"])
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    assert float((pred.view_as(rpred).cpu() - rpred).abs().max()) < 0.05
    assert abs(l1 - float(f["loss1"])) < 1e-2 and abs(l2 - float(f["loss2"])) < 1e-2, (l1, l2)
    gn = float(store.grad.double().norm())
    assert abs(gn - float(f["grad_norm"])) < 0.01 * float(f["grad_norm"]), (gn, float(f["grad_norm"]))
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    bad = []
    for k, v in ref.items():
        mine = float(store.g(k).double().norm())
        if float(v) > 1e-3 * float(f["grad_norm"]) and abs(mine - float(v)) > 0.05 * float(v):
            bad.append((k, mine, float(v)))
    assert not bad, bad[:10]
    sl = {"g_video_proj": ("video_model.proj", (slice(0, 8), slice(0, 16))),
          "g_head": ("pred_model.head.weight", (slice(None), slice(None))),
          "g_cfc11": ("video_model.transformer.resblocks.11.mlp.c_fc.weight", (slice(0, 8), slice(0, 16))),
          "g_temporal": ("video_model.temporal_embedding", (slice(None), slice(0, 16)))}
    for k, (name, idx) in sl.items():
        # slices of single gradient tensors against the reference's: the SURVEY 8d gate for a tensor is cosine >= 0.98, i.e. rel-L2
        # <= 0.2; these slices are held to 0.08 (the positional-embedding gradient, behind ln_pre's cancellation at the END of the
        # backward chain, measures 0.04 with the fp32 gradient chain -- the default again since round 4 -- and 0.096 with the opt-in
        # bf16 gradient stream; profiles/r03_bf16_streams_ab.txt)
        assert rel(store.g(name)[idx], torch.tensor(f[k])) < 0.08, (k, rel(store.g(name)[idx], torch.tensor(f[k])))


def test_h14_full_size_against_reference_golden(gpu, golden):
    """BASELINE config 3 architecture: the real TVTSv2_H_14 class (1.22 G parameters) ran in the build container at
    B=2, T=4; same synthetic parameters and batch here."""
    torch.set_num_threads(min(32, torch.get_num_threads()))
    f = golden("model_h14_cfg3")
    m, oarch, P = build(arch_name="H_14", seed=0)
    del P
    batch = O.synth_batch(oarch, B=2, T=4, seed=0)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    rte, rve, rpred = torch.tensor(f["te"]), torch.tensor(f["ve"]), torch.tensor(f["pred"])
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    assert float((pred.view_as(rpred).cpu() - rpred).abs().max()) < 0.05
    assert abs(l1 - float(f["loss1"])) < 1e-2 and abs(l2 - float(f["loss2"])) < 1e-2, (l1, l2)
    gn = float(store.grad.double().norm())
    assert abs(gn - float(f["grad_norm"])) < 0.02 * float(f["grad_norm"]), (gn, float(f["grad_norm"]))
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    bad = []
    for k, v in ref.items():
        mine = float(store.g(k).double().norm())
        if float(v) > 1e-3 * float(f["grad_norm"]) and abs(mine - float(v)) > 0.06 * float(v):
            bad.append((k, mine, float(v)))
    assert not bad, bad[:10]
    sl = {"g_video_proj": ("video_model.proj", (slice(0, 8), slice(0, 16))),
          "g_head": ("pred_model.head.weight", (slice(None), slice(0, 64))),
          "g_lnpost": ("video_model.ln_post.weight", (slice(None),)),
          "g_cfc31": ("video_model.transformer.resblocks.31.mlp.c_fc.weight", (slice(0, 8), slice(0, 16))),
          "g_temporal": ("video_model.temporal_embedding", (slice(None), slice(0, 16)))}
    for k, (name, idx) in sl.items():
        assert rel(store.g(name)[idx], torch.tensor(f[k])) < 0.1, (k, rel(store.g(name)[idx], torch.tensor(f[k])))
    assert rel(store.g("video_model.conv1.weight")[:4].reshape(4, -1), torch.tensor(f["g_conv"])) < 0.1


@pytest.mark.parametrize("fp8", [False, True])
def test_h14_t16_b2_against_oracle(gpu, fp8):
    """BASELINE configs[3] / [4] at their literal per-GPU batch on the REAL architecture: full-size ViT-H/14 (32 layers, width 1280,
    head dim 80, 1.22 G parameters), num_frames = 16 (the temporal table widened past the reference's 12 rows -- the reference
    class cannot run T > 12: video_encoder_ViT_H_14.py:419-484, model_dist_TVTSv2_ViT_H_14.py:65-66), tube mask 0.7, B = 2, 4 x
    32-token captions, against the fp32 CPU oracle run HERE on the same parameters and batch.  bf16: the SURVEY 8d gates (per-row
    cosine >= 0.9995, rel-L2 <= 2 %, |d loss| <= 1e-2, gradient norm 2 % as in the T = 4 reference golden, per-tensor cosine 0.995);
    e4m3 forward + input-gradient + weight-gradient GEMMs (second step after the calibration step): the e4m3 gates of
    test_h14_full_size_fp8_against_reference_golden."""
    import psutil
    if psutil.virtual_memory().available < 64 * 2 ** 30:
        pytest.skip("the fp32 CPU oracle's H/14 autograd graph at 2 x 16 frames needs ~40 GB of host memory")
    from tvts_amd import arch as A
    from tvts_amd.model._common import TVTSv2Base
    torch.set_num_threads(min(64, torch.get_num_threads()))
    a = dict(A.ARCHS["H_14"], num_frames=16)
    if fp8:
        a.update(fp8=True, fp8_dgrad=True, fp8_wgrad=True)
    oarch = dict(O.ARCHS["H_14"], num_frames=16)
    assert a["mask_ratio"] == oarch["mask_ratio"] == 0.7
    P = O.synth_params(oarch, seed=3)
    m = TVTSv2Base(ARGS, arch=a)
    m.load_state_dict(P, strict=True)
    batch = O.synth_batch(oarch, B=2, T=16, seed=4, caption_len=32)
    assert batch["video"].shape[1] == 16 and m.engine.prepare_batch(batch)["T"] == 16
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    del P
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    if fp8:
        m.engine.end_step()
        l1, l2, te, ve, pred, store = engine_step(m, batch)
        assert m.engine._f8_tensor_mode
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    if fp8:
        assert min_cos(ve, rve) > 0.995 and rel(ve, rve) < 0.1, (min_cos(ve, rve), rel(ve, rve))
        assert abs(l1 - r1) < 5e-2 and abs(l2 - r2) < 5e-2, (l1, r1, l2, r2)
        # per-tensor NORMS within 15 % as in the T = 4 reference golden.  Per-tensor cosines are reported and held to 0.9 only: with
        # two clips the 2 x 2 similarity at temperature 0.05 turns the e4m3 perturbation of the video embeddings (cosine 0.995) into a
        # different mix of the two clips' terms in every gradient that passes the contrastive loss -- the unquantised text tower's
        # tensors show it as much as the ViT's (measured worst: class_embedding 0.929, text resblocks.2 out_proj 0.952)
        tot, tot_ref, worst = check_grads(store, grads, gn_tol=0.05, cos_tol=0.9)
        bad = [(k, mine, ref) for c, k, mine, ref in worst if abs(mine - ref) > 0.15 * ref]
        assert not bad, bad[:10]
    else:
        assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
        assert float((pred.view_as(rpred).cpu() - rpred).abs().max()) < 0.05
        assert abs(l1 - r1) < 1e-2 and abs(l2 - r2) < 1e-2, (l1, r1, l2, r2)
        tot, tot_ref, worst = check_grads(store, grads, gn_tol=0.02, cos_tol=0.995)
        print(f"   [H/14 T=16 B=2] loss {l1:.5f}/{l2:.5f} (oracle {r1:.5f}/{r2:.5f}), ve min cos {min_cos(ve, rve):.6f} rel {rel(ve, rve):.4f}, "
              f"grad norm {tot / tot_ref - 1:+.3%}, worst tensor cosine {worst[0][0]:.5f} {worst[0][1]}")


@pytest.mark.parametrize("h14", [False, True])
def test_sort_head_used_rows_only(gpu, h14):
    """The sort head's last block evaluated on the rows the model reads (the NT transcript rows: sort_transformer.py:131-141) and
    the text tower's last block on the EOT rows (CLIP/clip/model.py:343-354) give the losses, embeddings, prediction and EVERY
    parameter gradient of the dense evaluation the reference performs, and both track the oracle."""
    from tvts_amd import arch as A
    m, oarch, P = build(arch=A.small_arch_h() if h14 else A.small_arch(), seed=11)
    batch = O.synth_batch(oarch, B=4, T=3, seed=12, caption_len=11)
    assert m.engine.sort_used_rows_only and m.engine.text_used_rows_only
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    g_used = store.grad.clone()
    m.engine.sort_used_rows_only = m.engine.text_used_rows_only = False
    d1, d2, dte, dve, dpred, store = engine_step(m, batch)
    g_dense = store.grad.clone()
    m.engine.sort_used_rows_only = m.engine.text_used_rows_only = True
    assert abs(l1 - d1) < 2e-3 and abs(l2 - d2) < 2e-3, (l1, d1, l2, d2)
    assert rel(te, dte) < 5e-3
    assert rel(pred, dpred) < 5e-3
    names = list(O.param_shapes(oarch).keys())
    gu, gd = g_used.double(), g_dense.double()
    assert abs(float(gu.norm()) - float(gd.norm())) < 2e-3 * float(gd.norm())
    assert float((gu * gd).sum() / (gu.norm() * gd.norm())) > 0.99995
    worst = []
    for k in names:
        if not (k.startswith("pred_model.") or k.startswith("text")):
            continue
        store.grad.copy_(g_used); u = store.g(k).double().clone()
        store.grad.copy_(g_dense); d = store.g(k).double().clone()
        if float(d.norm()) > 0:
            c = float((u * d).sum() / (u.norm() * d.norm() + 1e-30))
            worst.append((c, k, float(u.norm()), float(d.norm())))
    worst.sort()
    assert worst and worst[0][0] > 0.999, worst[:5]
    # and against the oracle
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    store.grad.copy_(g_used)
    assert abs(l2 - r2) < 1e-2 and rel(pred.view_as(rpred), rpred) < 0.03
    check_grads(store, grads)


@pytest.mark.parametrize("wgrad", [False, True])
def test_h14_full_size_fp8_against_reference_golden(gpu, golden, wgrad):
    """BASELINE config 4 / 5 on its own architecture: the full-size TVTSv2 ViT-H/14 (32 layers, width 1280, head dim 80, 1.22 G
    parameters) with the e4m3 forward and input-gradient GEMMs -- and (wgrad) the e4m3 weight gradients under per-tensor delayed
    scales, second step after the calibration step -- against the REFERENCE's fp32 outputs of the same parameters and
    batch (the golden of test_h14_full_size_against_reference_golden) at the fp8 tolerance: video embeddings cosine >= 0.995,
    losses within 5e-2, total gradient norm within 5 %, per-tensor gradient norms of every sizeable tensor within 15 %."""
    from tvts_amd import arch as A
    from tvts_amd.model._common import TVTSv2Base
    f = golden("model_h14_cfg3")
    a = dict(A.ARCHS["H_14"], fp8=True, fp8_dgrad=True, fp8_wgrad=wgrad)
    oarch = O.ARCHS["H_14"]
    P = O.synth_params(oarch, seed=0)
    m = TVTSv2Base(ARGS, arch=a)
    m.load_state_dict(P, strict=True)
    del P
    batch = O.synth_batch(oarch, B=2, T=4, seed=0)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    if wgrad:
        m.engine.end_step()
        l1, l2, te, ve, pred, store = engine_step(m, batch)
        assert m.engine._f8_tensor_mode and len(m.engine._f8_ids) == 12 * a["layers"]
    assert len(store.w8) == len(store.w8t) == 6 * a["layers"]
    rte, rve, rpred = torch.tensor(f["te"]), torch.tensor(f["ve"]), torch.tensor(f["pred"])
    assert min_cos(te, rte) > 0.9995                                  # the text tower is not quantised
    assert min_cos(ve, rve) > 0.995 and rel(ve, rve) < 0.1, (min_cos(ve, rve), rel(ve, rve))
    assert abs(l1 - float(f["loss1"])) < 5e-2 and abs(l2 - float(f["loss2"])) < 5e-2, (l1, l2)
    gn = float(store.grad.double().norm())
    assert abs(gn - float(f["grad_norm"])) < 0.05 * float(f["grad_norm"]), (gn, float(f["grad_norm"]))
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    bad = []
    for k, v in ref.items():
        mine = float(store.g(k).double().norm())
        if float(v) > 1e-3 * float(f["grad_norm"]) and abs(mine - float(v)) > 0.15 * float(v):
            bad.append((k, mine, float(v)))
    assert not bad, bad[:10]


def test_fp8_forward_path(gpu):
    """BASELINE config 4's weight / activation format on a small model: the six linear layers of every ViT block run their
    FORWARD product on per-tensor-scaled e4m3 copies (tvts_gemm_nt_fp8), the backward keeps the bf16 operands.  Checked
    against the oracle with the same quantise -> dequantise emulation (tight), and against the unquantised fp32 oracle
    (the fp8 tolerance: cosine >= 0.995, |d loss| <= 5e-2, gradient direction >= 0.97 per tensor, gradient norm within 3 %;
    every e4m3 gate of this file carries the margin measured on an MI355X beside it -- `pytest -s` prints the current ones)."""
    from tvts_amd import arch as A
    a = A.small_arch(fp8=True)
    m, oarch, P = build(arch=a, seed=3)
    batch = O.synth_batch(oarch, B=4, T=3, seed=5, caption_len=11)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)  # oarch carries fp8=True: emulated quantisation
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert len(store.w8) == 6 * a["layers"] and all(k.startswith("video_model.transformer.resblocks.") for k in store.w8)
    assert rel(te, rte) < 0.02  # text tower is not quantised
    assert min_cos(ve, rve) > 0.999 and rel(ve, rve) < 0.04, (min_cos(ve, rve), rel(ve, rve))
    assert abs(l1 - r1) < 2e-2 and abs(l2 - r2) < 2e-2, (l1, r1, l2, r2)
    # rounding decisions differ between the bf16-fed kernel and the fp32-fed emulation for values near an e4m3 tie; the
    # temperature-0.05 softmax amplifies that into the text tower's gradients
    # measured (round 5, MI355X): gradient norm +0.5 %, worst tensor cosine 0.989 (text_ln_final.bias), ViT tensors >= 0.991
    tot, tot_ref, worst = check_grads(store, grads, gn_tol=0.03, cos_tol=0.97)
    # and against the unquantised model: the price of e4m3
    oa32 = dict(oarch, fp8=False)
    f1, f2, fte, fve, fpred, fgrads = oracle_step(P, batch, oa32)
    assert min_cos(ve, fve) > 0.995 and abs(l1 - f1) < 5e-2 and abs(l2 - f2) < 5e-2, (min_cos(ve, fve), l1, f1, l2, f2)


def test_fp8_forward_path_h14_structure(gpu):
    """The same e4m3 forward path on BASELINE config 4's STRUCTURE: head dim 80, 14x14 patches, erf-GELU MLP (the fp8 GEMM's
    gelu + pre-activation epilogue), OpenCLIP block order, pooled tail; width 640 so that every quantised GEMM has K % 128 == 0
    like the real H/14 (1280 / 5120).  Tolerances as in test_fp8_forward_path."""
    from tvts_amd import arch as A
    a = A.small_arch_h(fp8=True, width=640, heads=8)
    m, oarch, P = build(arch=a, seed=4)
    batch = O.synth_batch(oarch, B=4, T=3, seed=6, caption_len=11)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    assert len(store.w8) == 6 * a["layers"]
    assert rel(te, rte) < 0.02
    assert min_cos(ve, rve) > 0.999 and rel(ve, rve) < 0.04, (min_cos(ve, rve), rel(ve, rve))
    assert abs(l1 - r1) < 2e-2 and abs(l2 - r2) < 2e-2, (l1, r1, l2, r2)
    # measured: gradient norm -0.8 %, worst tensor cosine 0.994 (block 1 mlp.c_proj.weight)
    check_grads(store, grads, gn_tol=0.03, cos_tol=0.97)
    f1, f2, fte, fve, fpred, fgrads = oracle_step(P, batch, dict(oarch, fp8=False))
    assert min_cos(ve, fve) > 0.995 and abs(l1 - f1) < 5e-2 and abs(l2 - f2) < 5e-2, (min_cos(ve, fve), l1, f1, l2, f2)


@pytest.mark.parametrize("h14", [False, True])
def test_fp8_dgrad_path(gpu, h14):
    """arch["fp8_dgrad"]: the input-gradient GEMMs of the ViT blocks' six linear layers on e4m3 operands too (output gradient one
    scale per token, the transposed weight's e4m3 copy, the MLP's activation-gradient gate in the fp8 kernel's epilogue); the
    weight gradients keep their bf16 operands.  Forward results are those of the forward-only fp8 path bit for bit; gradients
    are held to the fp8 tolerance against the oracle's emulation and stay aligned with the bf16-backward ones."""
    from tvts_amd import arch as A
    mk = (lambda **kw: A.small_arch_h(width=640, heads=8, **kw)) if h14 else A.small_arch
    m0, oarch, P = build(arch=mk(fp8=True), seed=4)
    batch = O.synth_batch(oarch, B=4, T=3, seed=6, caption_len=11)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store0 = engine_step(m0, batch)
    g0 = store0.grad.clone()
    m1, _, _ = build(arch=mk(fp8=True, fp8_dgrad=True), seed=4)
    k1, k2, te1, ve1, pred1, store = engine_step(m1, batch)
    assert len(store.w8t) == len(store.w8) == 6 * oarch["layers"]
    assert torch.equal(ve1, ve) and torch.equal(te1, te) and k1 == l1 and k2 == l2   # same forward
    # measured (B-style / H/14-style): gradient norm -0.3 % / -0.8 %, worst tensor cosine 0.991 (mlp.c_fc.weight) / 0.991 (ln_2.weight)
    check_grads(store, grads, gn_tol=0.03, cos_tol=0.97)
    g1 = store.grad
    cos = float(torch.nn.functional.cosine_similarity(g0.double().flatten(), g1.double().flatten(), dim=0))
    assert cos > 0.995 and abs(float(g1.double().norm()) / float(g0.double().norm()) - 1) < 0.02, cos
    assert not torch.equal(g0, g1)   # the e4m3 path did run


@pytest.mark.parametrize("h14", [False, True])
def test_fp8_wgrad_path(gpu, h14):
    """arch["fp8_wgrad"] (BASELINE config 5 in full): the weight gradients of the ViT blocks' six linear layers on e4m3 operands as
    well -- tvts_gemm_tn_fp8 (ds_read_b64_tr_b8 fragments, K = 128 scaled MFMA), every operand copy under ONE delayed scale per
    tensor, the bias gradients from the same bytes.  The first step is the calibration step: per-token copies and bf16 weight
    gradients (= the fp8_dgrad path, bit for bit) while the maxima are recorded; Engine.end_step() turns them into scales and the
    second step runs per tensor.  Its results are held to the fp8 tolerance against the oracle and stay aligned with the
    per-token / bf16-weight-gradient ones."""
    from tvts_amd import arch as A
    mk = (lambda **kw: A.small_arch_h(width=640, heads=8, **kw)) if h14 else A.small_arch
    m0, oarch, P = build(arch=mk(fp8=True, fp8_dgrad=True), seed=4)
    batch = O.synth_batch(oarch, B=4, T=3, seed=6, caption_len=11)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store0 = engine_step(m0, batch)
    g0 = store0.grad.clone()
    m1, _, _ = build(arch=mk(fp8_wgrad=True), seed=4)
    assert m1.engine.fp8_wgrad and m1.arch["fp8"] and m1.arch["fp8_dgrad"] and not m1.engine._f8_tensor_mode
    k1, k2, te1, ve1, pred1, store = engine_step(m1, batch)        # calibration step: the fp8_dgrad path
    assert torch.equal(ve1, ve) and torch.equal(te1, te) and torch.equal(store.grad, g0)
    m1.engine.end_step()
    assert m1.engine._f8_tensor_mode and len(m1.engine._f8_ids) == 12 * oarch["layers"]
    n = len(m1.engine._f8_ids)
    assert bool((m1.engine._f8_scale[:n] > 0).all()) and float(m1.engine._f8_amax.abs().max()) == 0.0
    q1, q2, te2, ve2, pred2, store = engine_step(m1, batch)        # per-tensor scales, e4m3 weight gradients
    assert min_cos(ve2, rve) > 0.999 and min_cos(te2, rte) > 0.9995 and abs(q1 - r1) < 5e-2 and abs(q2 - r2) < 5e-2, (q1, r1, q2, r2)
    # measured (B-style / H/14-style): gradient norm +1.2 % / -0.8 %, worst tensor cosine 0.984 / 0.987 (the MLP weights, whose
    # operands are e4m3 on both sides of all three products): the gates sit 1.5-2 % under what is measured, not at 0.9
    check_grads(store, grads, gn_tol=0.03, cos_tol=0.97)
    g2 = store.grad.clone()
    cos = float(torch.nn.functional.cosine_similarity(g0.double().flatten(), g2.double().flatten(), dim=0))
    assert cos > 0.99 and abs(float(g2.double().norm()) / float(g0.double().norm()) - 1) < 0.03, cos
    assert not torch.equal(g0, g2)
    # the bias gradients come from the e4m3 bytes of the output gradient: aligned with the bf16 column sums
    for l in range(oarch["layers"]):
        for nm in ("attn.qkv.bias", "mlp.c_fc.bias", "mlp.c_proj.bias"):
            k = f"video_model.transformer.resblocks.{l}.{nm}"
            a_, b_ = store.g(k).double().flatten().cpu(), grads[k].double().flatten()
            assert float(torch.nn.functional.cosine_similarity(a_, b_, dim=0)) > 0.97, k
    # the attention outputs' / attention input gradients' copies come out of the divided-attention kernels themselves (default);
    # with the quantiser passes instead (fp8_attn_copies = False): the same bytes, the same maxima, the same step
    m4, _, _ = build(arch=mk(fp8_wgrad=True, fp8_attn_copies=False), seed=4)
    engine_step(m4, batch); m4.engine.end_step()
    n4 = len(m4.engine._f8_ids)
    ids = {k: (m1.engine._f8_ids[k], m4.engine._f8_ids[k]) for k in m4.engine._f8_ids}
    _, _, _, ve4, _, store4 = engine_step(m4, batch)
    assert torch.equal(ve4, ve2) and torch.equal(store4.grad, g2)
    for k, (i1, i4) in ids.items():
        assert float(m1.engine._f8_amax[i1]) == float(m4.engine._f8_amax[i4]), k
    assert n4 == n
    # the MLP's two operand copies written by the producing GEMMs' epilogues instead of quantiser passes (opt-in): the same bytes
    m3, _, _ = build(arch=mk(fp8_wgrad=True, fp8_epilogue_copies=True), seed=4)
    engine_step(m3, batch); m3.engine.end_step()
    _, _, _, ve3, _, store3 = engine_step(m3, batch)
    assert torch.equal(ve3, ve2) and torch.equal(store3.grad, g2)
    m1.engine.end_step()
    _, _, _, _, _, store = engine_step(m1, batch)                  # scales = the previous step's maxima
    g3 = store.grad.clone()
    _, _, _, _, _, store = engine_step(m1, batch)                  # no end_step in between: the same scales
    assert torch.equal(g3, store.grad)                             # identical scales, identical steps: identical bits


def test_b16_config2_against_reference_golden(gpu, golden):
    """The headline architecture (BASELINE config 2's model): the real TVTSv2_B_16 class ran in the build container at
    B=2, T=4 with tube mask 0.5 (98 of 196 patches kept: the fused SPACE / TIME attention kernels' shapes)."""
    f = golden("model_b16_cfg2")
    m, oarch, P = build(arch_name="B_16", seed=0)
    del P
    batch = O.synth_batch(oarch, B=2, T=4, seed=int(f["batch_seed"]))
    l1, l2, te, ve, pred, store = engine_step(m, batch)
    rte, rve, rpred = torch.tensor(f["te"]), torch.tensor(f["ve"]), torch.tensor(f["pred"])
    assert min_cos(te, rte) > 0.9995 and rel(te, rte) < 0.02, (min_cos(te, rte), rel(te, rte))
    assert min_cos(ve, rve) > 0.9995 and rel(ve, rve) < 0.02, (min_cos(ve, rve), rel(ve, rve))
    assert float((pred.view_as(rpred).cpu() - rpred).abs().max()) < 0.05
    assert abs(l1 - float(f["loss1"])) < 1e-2 and abs(l2 - float(f["loss2"])) < 1e-2, (l1, l2)
    gn = float(store.grad.double().norm())
    assert abs(gn - float(f["grad_norm"])) < 0.01 * float(f["grad_norm"]), (gn, float(f["grad_norm"]))
    ref = dict(zip([str(s) for s in f["gn_names"]], f["gn_vals"]))
    bad = []
    for k, v in ref.items():
        mine = float(store.g(k).double().norm())
        if float(v) > 1e-3 * float(f["grad_norm"]) and abs(mine - float(v)) > 0.05 * float(v):
            bad.append((k, mine, float(v)))
    assert not bad, bad[:10]
    for k, (name, idx) in {"g_video_proj": ("video_model.proj", (slice(0, 8), slice(0, 16))),
                           "g_head": ("pred_model.head.weight", (slice(None), slice(None))),
                           "g_pos": ("video_model.positional_embedding", (slice(None), slice(0, 16)))}.items():
        # slices of single gradient tensors against the reference's: the SURVEY 8d gate for a tensor is cosine >= 0.98, i.e. rel-L2
        # <= 0.2; these slices are held to 0.08 (the positional-embedding gradient, behind ln_pre's cancellation at the END of the
        # backward chain, measures 0.04 with the fp32 gradient chain -- the default again since round 4 -- and 0.096 with the opt-in
        # bf16 gradient stream; profiles/r03_bf16_streams_ab.txt)
        assert rel(store.g(name)[idx], torch.tensor(f[k])) < 0.08, (k, rel(store.g(name)[idx], torch.tensor(f[k])))


@pytest.mark.parametrize("h14", [False, True])
def test_residual_stream_precision_options(gpu, h14):
    """Default since round 5: the HYBRID stream -- the residual stream of the space-time blocks and its gradient in bf16, the CLS
    token's row of every clip in fp32 side arrays (arch["hybrid_stream"], experiments/dbg/bf16_residual_rows.py says why that one row).
    arch["hybrid_stream"] = False: both streams fp32 on every row (rounds 1, 2, 4).  arch["bf16_grad_stream"] / arch["bf16_residual"]
    = True: the gradient / both streams bf16 on EVERY row (round 3; the forward variant fails the |d loss| <= 1e-2 gate on a 3-pair toy).
    All hold the SURVEY 8d gradient gates against the oracle; the hybrid stream's forward sits with the fp32 streams', not with the
    all-bf16 stream's."""
    from tvts_amd import arch as A
    mk = (lambda **kw: A.small_arch_h(**kw)) if h14 else A.small_arch
    m0, oarch, P = build(arch=mk(bf16_grad_stream=True), seed=4)
    batch = O.synth_batch(oarch, B=4, T=3, seed=6, caption_len=11)
    r1, r2, rte, rve, rpred, grads = oracle_step(P, batch, oarch)
    l1, l2, te, ve, pred, store0 = engine_step(m0, batch)
    assert m0.engine.bf16_grad_stream and not m0.engine.bf16_residual and not m0.engine.cls32
    assert m0.engine.buf["vit.x1"].dtype == torch.float32 and "vit.s.dsres" not in m0.engine.buf
    check_grads(store0, grads, cos_tol=0.99)   # the all-row bf16 streams: measured >= 0.998 on B/16, gate one notch under the default's
    g0 = store0.grad.clone()
    m2, _, _ = build(arch=mk(hybrid_stream=False), seed=4)       # both streams fp32
    j1, j2, te2, ve2, pred2, store2 = engine_step(m2, batch)
    assert "vit.s.dsres" in m2.engine.buf and not m2.engine.bf16_grad_stream and not m2.engine.cls32
    check_grads(store2, grads)
    assert torch.equal(ve2, ve) and torch.equal(te2, te) and j1 == l1 and j2 == l2   # same forward
    m1, _, _ = build(arch=mk(bf16_residual=True), seed=4)           # both streams bf16 on every row
    k1, k2, te1, ve1, pred1, store1 = engine_step(m1, batch)
    assert m1.engine.bf16_grad_stream and not m1.engine.cls32
    assert m1.engine.buf["vit.x1"].dtype == torch.bfloat16 and m1.engine.buf["vit0.s_res"].dtype == torch.bfloat16
    check_grads(store1, grads, cos_tol=0.99)
    assert min_cos(ve1, rve) > 0.9995 and rel(ve1, rve) < 0.02 and abs(k1 - r1) < 2e-2 and abs(k2 - r2) < 1e-2
    assert min_cos(ve1, ve) > 0.9999 and not torch.equal(ve1, ve)
    for ga in (g0, store1.grad):
        cos = float(torch.nn.functional.cosine_similarity(ga.double().flatten(), store2.grad.double().flatten(), dim=0))
        assert cos > 0.999 and not torch.equal(ga, store2.grad), cos
    m3, _, _ = build(arch=mk(), seed=4)                             # the default: hybrid
    h1, h2, te3, ve3, pred3, store3 = engine_step(m3, batch)
    e = m3.engine
    assert e.cls32 and e.bf16_residual and e.bf16_grad_stream
    assert e.buf["vit.x1"].dtype == torch.bfloat16 and e.buf["vit.xc1"].dtype == torch.float32 and e.buf["vit.xc1"].shape[0] == 4
    check_grads(store3, grads)                                       # the default gates (cosine 0.995, gradient norm 1 %)
    assert min_cos(ve3, rve) > 0.9995 and rel(ve3, rve) < 0.02 and abs(h1 - r1) < 1e-2 and abs(h2 - r2) < 1e-2
    # the video embedding is read from the CLS rows: the hybrid stream's distance from the oracle is the fp32 streams' (the bf16
    # GEMM operands' noise), the all-row bf16 stream's is visibly larger
    d_fp32, d_hyb, d_bf16 = rel(ve2, rve), rel(ve3, rve), rel(ve1, rve)
    print(f"video embedding rel-L2 vs oracle: fp32 streams {d_fp32:.2e}, hybrid {d_hyb:.2e}, bf16 on every row {d_bf16:.2e}; "
          f"|d loss1| {abs(j1 - r1):.2e} / {abs(h1 - r1):.2e} / {abs(k1 - r1):.2e}")
    assert d_hyb < 1.25 * d_fp32 + 2e-4 and d_hyb < d_bf16
    gcos = lambda st: float(torch.nn.functional.cosine_similarity(st.grad.double().flatten(), store2.grad.double().flatten(), dim=0))
    assert gcos(store3) > 0.9995
