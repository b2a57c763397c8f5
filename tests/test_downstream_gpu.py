"""Inference-only model copies (v2/downstream/model_TVTSv2_ViT_B_16.py) on the HIP engine against the
reference's own outputs (tests/golden/downstream_b16.npz) and the zero-shot arithmetic against torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tvts_oracle as O  # noqa: E402  (checker only)

DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tvts_amd.downstream.model_TVTSv2_ViT_B_16 import TVTSv2_B_16
    m = TVTSv2_B_16(load_checkpoint=None, pretrained=False)
    arch = dict(O.ARCHS["B_16"], mask_ratio=0.0, sort_head=False)
    P = O.synth_params(arch, seed=0)
    assert list(m.state_dict().keys()) == list(P.keys())  # no pred_model.* keys, reference order
    m.load_state_dict(P, strict=True)
    return m


def test_downstream_b16_against_reference_golden(model, golden):
    f = golden("downstream_b16")
    b = O.synth_batch(O.ARCHS["B_16"], B=2, T=4, seed=int(f["batch_seed"]), n_trans=1)
    data = {"text": b["text"], "video": b["video"], "keep_ind": torch.arange(196).unsqueeze(0).expand(2, -1)}
    te, ve = model(data, return_embeds=True)
    assert rel(te, f["te"]) < 0.02 and rel(ve, f["ve"]) < 0.02, (rel(te, f["te"]), rel(ve, f["ve"]))
    cos = torch.nn.functional.cosine_similarity(ve.cpu().double(), torch.tensor(f["ve"]).double(), dim=1)
    assert float(cos.min()) > 0.9995
    sims = model(data, return_embeds=False)
    assert float((sims.cpu() - torch.tensor(f["sims"])).abs().max()) < 5e-3
    assert all(not p.requires_grad for p in model.parameters())


def test_zero_shot_text_pass_and_logits(model, golden):
    """zero_recognition_TVTSv2_ViT_B_16.py:67-100: prompts beside dummy single frames with ONE tube-mask row for the batch."""
    from tvts_amd.downstream import zero_shot as Z
    f = golden("downstream_b16")
    prompts = torch.tensor(f["prompts"])
    data = {"text": prompts, "video": torch.zeros(3, 3, 224, 224), "keep_ind": torch.arange(196).unsqueeze(0)}
    cls_emb, _ = model(data, return_embeds=True)
    assert rel(cls_emb, f["cls_emb"]) < 0.02
    w = Z.class_embedding(model, prompts, 196)
    ref = torch.tensor(f["cls_emb"])
    ref = ref / ref.norm(dim=-1, keepdim=True)
    ref = ref.mean(0)
    ref = ref / ref.norm()
    assert rel(w, ref) < 0.02
    g = torch.Generator().manual_seed(1)
    v = torch.randn(7, 512, generator=g).to(DEV)
    W = torch.randn(512, 11, generator=g).to(DEV)
    logits = Z.class_logits(v, W)
    want = 100.0 * (v / v.norm(dim=-1, keepdim=True)) @ W
    assert rel(logits, want.cpu()) < 1e-5
    target = want.argmax(1)
    assert Z.accuracy(logits, target, topk=(1, 5)) == [7.0, 7.0]
    assert Z.accuracy(logits, want.argmin(1), topk=(1, 11)) == [0.0, 7.0]


@pytest.mark.parametrize("name,B,T,n", [("B_32", 2, 5, 49), ("H_14", 1, 2, 256)])
def test_downstream_other_archs_against_reference_golden(name, B, T, n, golden):
    """TVTSv2_B_32 (49 unmasked patches: fused SPACE kernels) and the full-size TVTSv2_H_14 (256 patches, head dim 80, pooled
    tail) of v2/downstream, against the reference's own outputs."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib
    mod = importlib.import_module(f"tvts_amd.downstream.model_TVTSv2_ViT_{name}")
    m = getattr(mod, f"TVTSv2_{name}")(load_checkpoint=None, pretrained=False)
    arch = dict(O.ARCHS[name], mask_ratio=0.0, sort_head=False)
    m.load_state_dict(O.synth_params(arch, seed=0), strict=True)
    f = golden("downstream_" + name.lower().replace("_", ""))
    b = O.synth_batch(O.ARCHS[name], B=B, T=T, seed=int(f["batch_seed"]), n_trans=1)
    te, ve = m({"text": b["text"], "video": b["video"], "keep_ind": torch.arange(n).unsqueeze(0).expand(B, -1)})
    assert rel(te, f["te"]) < 0.02 and rel(ve, f["ve"]) < 0.02, (rel(te, f["te"]), rel(ve, f["ve"]))
    cos = torch.nn.functional.cosine_similarity(ve.cpu().double(), torch.tensor(f["ve"]).double(), dim=1)
    assert float(cos.min()) > 0.9995
    del m
    torch.cuda.empty_cache()
