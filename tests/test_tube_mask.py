"""Tube mask drawn on the device.  The reference draws np.random.shuffle(arange(ppf))[:n_keep] per sample
in the dataset worker (v2/data_loader/YTTemporal_dataset.py:207-213); the worker's random stream is not a contract, the
distribution is: an unsorted n_keep-prefix of a uniformly random permutation.  CPU tests pin the oracle's restatement of the
device draw to that contract; GPU tests hold the kernel bit-exact to the oracle and the step on a device-drawn mask equal to
the step on the same mask passed in as `keep_ind`."""
import types

import numpy as np
import pytest
import torch

from oracle import tvts_oracle as O  # checker only

DEV = "cuda:0"
CASES = [(49, 49), (49, 24), (196, 98), (256, 76), (256, 256), (1024, 307), (5, 1)]


@pytest.mark.parametrize("ppf,n", CASES)
def test_oracle_draw_is_a_permutation_prefix(ppf, n):
    m = O.tube_mask(7, 100, 4, ppf, n)
    assert m.shape == (4, n) and m.dtype == np.int32
    for row in m:
        assert len(set(row.tolist())) == n and row.min() >= 0 and row.max() < ppf
    # reproducible per (seed, sample number), independent of how samples are batched
    again = np.concatenate([O.tube_mask(7, 100, 1, ppf, n), O.tube_mask(7, 101, 3, ppf, n)])
    assert np.array_equal(m, again)
    if n < ppf:
        assert not np.array_equal(m[0], m[1]) or ppf < 4
        assert not np.array_equal(O.tube_mask(8, 100, 1, ppf, n)[0], m[0]) or ppf < 4
    # a prefix of the same permutation: fewer kept patches = the first entries of the longer draw
    assert np.array_equal(O.tube_mask(7, 100, 4, ppf, max(n - 1, 0)), m[:, :max(n - 1, 0)])


def test_oracle_draw_is_uniform():
    # marginal: every patch is kept with probability n/ppf; first position uniform over patches (chi-square, 99.9 % bound)
    ppf, n, S = 49, 24, 4000
    m = O.tube_mask(3, 0, S, ppf, n)
    kept = np.bincount(m.reshape(-1), minlength=ppf)
    exp = S * n / ppf
    var = S * (n / ppf) * (1 - n / ppf)
    assert np.abs(kept - exp).max() < 4.5 * np.sqrt(var)
    first = np.bincount(m[:, 0], minlength=ppf)
    chi2 = ((first - S / ppf) ** 2 / (S / ppf)).sum()
    assert chi2 < 90.0  # 48 dof: P(chi2 > 90) ~ 2e-4
    # the prefix is unsorted (a shuffled arange is), not a sorted index list
    assert (np.diff(m, axis=1) < 0).any(axis=1).mean() > 0.99


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("ppf,n", CASES)
def test_device_draw_bit_exact(gpu, ppf, n):
    from tvts_amd import hip as K
    for seed, first, B in ((7, 100, 4), (2 ** 63 + 5, 2 ** 40, 3), (0, 0, 1)):
        got = K.tube_mask(seed, first, B, ppf, n, device=DEV)
        assert got.dtype == torch.int32 and tuple(got.shape) == (B, n)
        assert np.array_equal(got.cpu().numpy(), O.tube_mask(seed, first, B, ppf, n))


@pytest.mark.gpu
def test_device_draw_edge_cases(gpu):
    from tvts_amd import hip as K
    assert K.tube_mask(1, 0, 0, 196, 98, device=DEV).shape == (0, 98)          # empty batch
    assert K.tube_mask(1, 0, 3, 196, 0, device=DEV).shape == (3, 0)           # nothing kept
    with pytest.raises(K.HipError):
        K.tube_mask(1, 0, 2, 2000, 10, device=DEV)                            # more patches than the LDS sort holds
    with pytest.raises(K.HipError):
        K.tube_mask(1, 0, 2, 49, 50, device=DEV)                              # n_keep > ppf
    big = K.tube_mask(11, 0, 4096, 196, 98, device=DEV).cpu().numpy()          # a whole node's batch in one launch
    assert np.array_equal(big[[0, 1777, 4095]], np.concatenate([O.tube_mask(11, i, 1, 196, 98) for i in (0, 1777, 4095)]))
    assert all(len(set(r.tolist())) == 98 for r in big[::97])


@pytest.mark.gpu
def test_step_with_device_mask_equals_step_with_that_mask_passed_in(gpu):
    from tvts_amd import arch as A
    from tvts_amd.model._common import TVTSv2Base
    a = A.small_arch()
    oarch = O.tiny_arch(**a)
    m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
    m.load_state_dict(O.synth_params(oarch, seed=2), strict=True)
    b = O.synth_batch(oarch, B=3, T=2, seed=4, caption_len=9)
    keep = torch.tensor(O.tube_mask(21, 9, 3, A.patches_per_frame(a), A.n_keep(a))).long()
    nb = {k: v for k, v in b.items() if k != "keep_ind"}
    with torch.no_grad():
        te0, ve0, pr0 = m(dict(nb, mask_seed=21, sample_offset=9))
        te1, ve1, pr1 = m(dict(nb, keep_ind=keep))
    assert torch.equal(ve0, ve1) and torch.equal(te0, te1) and torch.equal(pr0, pr1)
    with pytest.raises(KeyError):
        m(nb)
