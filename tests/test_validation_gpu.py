"""Validation step on the GPU (v2/trainer/trainer.py:527-635, v2/model/metric.py): rank kernel and metric
functions against the reference's own results (tests/golden/metrics.npz) and the oracle, and the trainer's _valid_epoch
end to end against the oracle's forward."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tvts_oracle as O  # noqa: E402  (checker only)

DEV = "cuda:0"


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return True


def test_rank_kernel_and_metrics_against_reference_golden(gpu, golden):
    from tvts_amd import hip as K
    from tvts_amd.model import metric as M
    f = golden("metrics")
    keys = [str(k) for k in f["keys"]]
    for name in ("rand_square", "ties_square", "two_caps", "two_caps_ties"):
        sims = torch.tensor(f["sims_" + name], device=DEV)
        assert np.array_equal(K.retrieval_ranks(sims, "t2v").cpu().numpy().astype(np.float64), O.t2v_ranks(f["sims_" + name]))
        assert np.array_equal(K.retrieval_ranks(sims, "v2t").cpu().numpy().astype(np.float64), O.v2t_ranks(f["sims_" + name]))
        for fn in ("t2v_metrics", "v2t_metrics"):
            got = getattr(M, fn)(sims)
            want = dict(zip(keys, f[f"{fn}_{name}"]))
            assert list(got.keys()) == keys
            for k in keys:
                assert abs(float(got[k]) - want[k]) < 1e-6 * max(1.0, abs(want[k])), (name, fn, k, got[k], want[k])


def test_metrics_with_query_masks(gpu, golden):
    """zero_ret_*.py hands the metrics MSRVTT's caption mask (metric.py:104-111,160-177)."""
    from tvts_amd.model import metric as M
    f = golden("metrics")
    keys = [str(k) for k in f["keys"]]
    sims, mask = torch.tensor(f["sims_two_caps"], device=DEV), f["query_mask_two_caps"]
    for fn in ("t2v_metrics", "v2t_metrics"):
        got = getattr(M, fn)(sims, query_masks=mask)
        want = dict(zip(keys, f[f"{fn}_two_caps_masked"]))
        for k in keys:
            assert abs(float(got[k]) - want[k]) < 1e-6 * max(1.0, abs(want[k])), (fn, k, got[k], want[k])


def test_rank_kernel_large_with_ties(gpu):
    from tvts_amd import hip as K
    g = torch.Generator().manual_seed(3)
    sims = (torch.randn(1536, 512, generator=g) * 3).round() / 4  # 3 captions per video, many exact ties
    d = sims.to(DEV)
    assert np.array_equal(K.retrieval_ranks(d, "t2v").cpu().numpy().astype(np.float64), O.t2v_ranks(sims.numpy()))
    assert np.array_equal(K.retrieval_ranks(d, "v2t").cpu().numpy().astype(np.float64), O.v2t_ranks(sims.numpy()))
    # a strided view (the trainer hands over a contiguous matrix, the ABI takes a leading dimension)
    wide = torch.zeros(1536, 640, device=DEV)
    wide[:, :512] = d
    assert torch.equal(K.retrieval_ranks(wide[:, :512], "v2t"), K.retrieval_ranks(d, "v2t"))


def test_valid_epoch_matches_oracle(gpu, capsys):
    from tvts_amd import arch as A
    from tvts_amd.model import metric as M
    from tvts_amd.model._common import TVTSv2Base
    from tvts_amd.trainer.trainer import Trainer_TVTSv2_B_16
    a = A.small_arch()
    oarch = O.tiny_arch(**a)
    P = O.synth_params(oarch, seed=9)
    args = types.SimpleNamespace(local_rank=0, rank=0, world_size=1)
    m = TVTSv2Base(args, arch=a)
    m.load_state_dict(P, strict=True)
    batches = [O.synth_batch(oarch, B=4, T=3, seed=70 + i, caption_len=9) for i in range(3)]

    class Loader(list):
        dataset_name = "YTVal"
    seen = {}

    def t2v_metrics(sims):
        seen["sims"] = sims.detach().clone()
        return M.t2v_metrics(sims)
    tr = object.__new__(Trainer_TVTSv2_B_16)  # the validation step only needs these attributes
    tr.model, tr.args, tr.valid_data_loader, tr.metrics = m, args, [Loader(batches)], [t2v_metrics, M.v2t_metrics]
    tr.writer, tr.tokenizer = None, None
    res = tr._valid_epoch(3)
    out = capsys.readouterr().out
    assert "[t2v_metrics]YTVal epoch 3, R@1:" in out and "[v2t_metrics]YTVal epoch 3" in out
    assert "Top-1 Accuracy for Frame Prediction:" in out

    # oracle side: fp32 forward of the same batches
    tes, ves, hits, count = [], [], 0, 0
    preds_engine = []
    with torch.no_grad():
        for b in batches:
            te, ve, pred = O.model_forward(P, b, oarch)
            tes.append(te); ves.append(ve)
            _, _, pe = m(b, return_embeds=True)
            preds_engine.append(pe.argmax(-1).cpu().numpy())
    ref_sims = O.sim_matrix(torch.cat(tes), torch.cat(ves))
    sims = seen["sims"].cpu()
    assert sims.shape == (12, 12)
    assert float((sims - ref_sims).abs().max()) < 2e-2, float((sims - ref_sims).abs().max())
    # metrics are exactly the reference's formulas applied to the similarity matrix the step produced
    for name, ranks in (("t2v_metrics", O.t2v_ranks), ("v2t_metrics", O.v2t_ranks)):
        want = O.cols2metrics(ranks(sims.numpy()))
        got = tr.last_val_metrics[0][name]
        for k, v in want.items():
            assert abs(float(got[k]) - v) < 1e-6 * max(1.0, abs(v)), (name, k, got[k], v)
    # sorting accuracy: a sample counts only when all NT positions are right
    h, c = 0, 0
    for b, pa in zip(batches, preds_engine):
        hh, cc = O.sorting_accuracy(pa, b["label"].numpy())
        h += hh; c += cc
    assert c == 12 and abs(res["val_loss_0"] - h / c) < 1e-12
