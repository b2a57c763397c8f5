"""A13: constructor-time initialisation from CLIP / OpenCLIP weights (v2/model/model_dist_TVTSv2_ViT_B_16.py:19-45,
..._B_32.py, ..._H_14.py:44-83).  The fixtures (tests/golden/make_golden.py::gen_ctor_init) hold, for every tensor of a freshly
constructed REAL reference model, the CRC-32 of its bytes -- the pretrained model being the reference's own CLIP / OpenCLIP
classes filled with oracle.synth_clip_state_dict(layout, 77).  The same seeded state dict goes through this package's mapping
(no GPU) and through its constructor (GPU): CLIP-derived tensors must come out bit-identical, timeattn / ln_3 at the
reference's constants, temporal_embedding and the sorting head fresh with the reference's spread."""
import types

import numpy as np
import pytest
import torch

from oracle import tvts_oracle as O
from tvts_amd import arch as A
from tvts_amd.model import clip_init as CI

ARCH_OF = {"b16": "B_16", "b32": "B_32", "h14": "H_14"}


def _layout_of(f):
    return {str(k): tuple(int(x) for x in str(s).split(",") if x) for k, s in zip(f["clip_keys"], f["clip_shapes"])}


@pytest.mark.parametrize("tag", ["b16", "b32", "h14"])
def test_mapping_matches_the_reference_constructor(golden, tag):
    f = golden("ctor_init_" + tag)
    arch = A.ARCHS[ARCH_OF[tag]]
    ref_layout = _layout_of(f)
    mine = CI.clip_layout(arch)
    # every key the product consumes exists in the real (Open)CLIP state dict with that shape; what it does not consume are the
    # scalars / buffers the reference never reads either
    for k, shp in mine.items():
        assert ref_layout[k] == tuple(shp), k
    unused = sorted(set(ref_layout) - set(mine))
    assert all(k in ("logit_scale", "input_resolution", "context_length", "vocab_size", "attn_mask") for k in unused), unused
    clip_sd = O.synth_clip_state_dict(ref_layout, int(f["seed"]))
    conv = CI.convert_clip_state_dict(clip_sd, arch)
    names = [str(n) for n in f["names"]]
    assert names == list(A.param_shapes(arch)), "state-dict order of the reference model"
    crc = dict(zip(names, (int(c) for c in f["crc"])))
    kind = dict(zip(names, (int(k) for k in f["kind"])))
    for n in names:
        if kind[n] == 0:
            assert n in conv and O.tensor_crc(conv[n]) == crc[n], n   # bit-identical to what the reference constructor left there
        else:
            assert n not in conv, n
    assert sorted(CI.fresh_names(arch)) == sorted(n for n in names if kind[n] != 0)
    # the rule of SURVEY App. B #4: fresh = temporal attention, its LayerNorm, the temporal table, the sorting head
    for n in names:
        fresh = "timeattn" in n or ".ln_3." in n or n.endswith("temporal_embedding") or n.startswith("pred_model.")
        assert (kind[n] != 0) == fresh, n


def test_mapping_rejects_wrong_shapes_and_missing_text_keys():
    arch = A.small_arch()
    sd = O.synth_clip_state_dict(CI.clip_layout(arch), 1)
    CI.convert_clip_state_dict(sd, arch)
    bad = dict(sd)
    bad["visual.conv1.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        CI.convert_clip_state_dict(bad, arch)
    extra = dict(sd)
    extra["visual.attnpool.weight"] = torch.zeros(2)  # strict=False: unexpected image-tower keys are dropped
    assert "video_model.attnpool.weight" not in CI.convert_clip_state_dict(extra, arch)
    miss = dict(sd)
    del miss["ln_final.weight"]
    with pytest.raises(KeyError):
        CI.convert_clip_state_dict(miss, arch)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["b16", "b32", "h14"])
def test_constructor_initialises_like_the_reference(golden, tag):
    from tvts_amd.model.model_dist_TVTSv2_ViT_B_16 import TVTSv2_B_16
    from tvts_amd.model.model_dist_TVTSv2_ViT_B_32 import TVTSv2_B_32
    from tvts_amd.model.model_dist_TVTSv2_ViT_H_14 import TVTSv2_H_14
    f = golden("ctor_init_" + tag)
    cls = {"b16": TVTSv2_B_16, "b32": TVTSv2_B_32, "h14": TVTSv2_H_14}[tag]
    clip_sd = O.synth_clip_state_dict(_layout_of(f), int(f["seed"]))
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    m = cls(args, load_checkpoint="", pretrained=clip_sd)
    assert m.initialised_from == "clip"
    sd = m.state_dict()
    names = [str(n) for n in f["names"]]
    assert list(sd) == names
    W = m.arch["width"]
    for n, crc, kind, mean, std, cv in zip(names, f["crc"], f["kind"], f["mean"], f["std"], f["const_value"]):
        t = sd[n].float().cpu()
        if int(kind) == 0:
            assert O.tensor_crc(t) == int(crc), n
        elif int(kind) == 1:   # timeattn zeros / ones, ln_3 ones / zeros, type_embed zeros, fresh biases of zero ...
            if "timeattn" in n or ".ln_3." in n or n.endswith("type_embed") or "norm" in n:
                assert bool((t == float(cv)).all()), (n, float(cv))
        else:                  # random in the reference too: same spread, not the same draw (a 4-element bias has no spread to compare)
            assert torch.isfinite(t).all() and float(t.abs().max()) <= 8 * float(std) + abs(float(mean)) + 1e-3, n
            if t.numel() >= 4096:
                assert abs(float(t.std()) - float(std)) < 0.1 * float(std) + 1e-4, (n, float(t.std()), float(std))
                assert abs(float(t.mean()) - float(mean)) < 0.2 * float(std) + 1e-4, n
    te = sd["video_model.temporal_embedding"].float()
    assert abs(float(te.std()) - W ** -0.5) < 0.1 * W ** -0.5
    # and the named class refuses to start from random weights silently when the pretrained model is not there
    with pytest.raises(RuntimeError, match="pretrained"):
        cls(args, load_checkpoint="")
    m2 = cls(args, load_checkpoint="", pretrained=False)
    assert m2.initialised_from == "random"
    del m, m2
    # the downstream inference classes initialise the same way (v2/downstream/model_TVTSv2_ViT_B_16.py:15-41; no sorting head there)
    import importlib
    dcls = getattr(importlib.import_module(f"tvts_amd.downstream.model_TVTSv2_ViT_{ARCH_OF[tag]}"), f"TVTSv2_{ARCH_OF[tag]}")
    d = dcls(pretrained=clip_sd)
    dsd = d.state_dict()
    for n, crc, kind in zip(names, f["crc"], f["kind"]):
        if n.startswith("pred_model."):
            assert n not in dsd
        elif int(kind) == 0 and n != "video_model.temporal_embedding":
            assert O.tensor_crc(dsd[n].float().cpu()) == int(crc), n
    with pytest.raises(RuntimeError, match="pretrained"):
        dcls()
