"""uint8 wire format of the video input: crop + ClipToTensor + Normalize fused into the tube-mask gather.
The kernel is held bit-exact to the reference's transform tail (tests/golden/transform.npz via the oracle), and the step on
uint8 frames must equal the step on the fp32 clip those frames normalise to."""
import types

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


@pytest.mark.parametrize("patch,img,H0,W0,use_crop", [(16, 64, 80, 72, False), (16, 64, 77, 91, True), (14, 56, 67, 60, True)])
def test_patch_gather_u8_bit_exact(gpu, golden, patch, img, H0, W0, use_crop):
    from tvts_amd import hip as K
    g = torch.Generator().manual_seed(patch + H0)
    B, T = 3, 2
    frames = torch.randint(0, 256, (B, T, H0, W0, 3), generator=g, dtype=torch.uint8)
    crop = torch.stack([torch.randint(0, H0 - img + 1, (B,), generator=g), torch.randint(0, W0 - img + 1, (B,), generator=g)], 1) \
        if use_crop else None
    gsz = img // patch
    n = 5
    keep = torch.stack([torch.randperm(gsz * gsz, generator=g)[:n].sort().values for _ in range(B)])
    Kc = 3 * patch * patch
    Kp = -(-Kc // 64) * 64
    cols = torch.full((B * T * n, Kp), 9.0, dtype=torch.bfloat16, device=DEV)
    K.patch_gather_u8(frames.to(DEV), keep.to(torch.int32).to(DEV), cols, B=B, T=T, n=n, img=img, patch=patch,
                      crop=None if crop is None else crop.to(torch.int32).to(DEV))
    video = O.frames_to_video(frames, img, crop)  # pinned bit-exact to the reference by tests/test_oracle_golden.py
    pix = video.reshape(B, T, 3, gsz, patch, gsz, patch).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, gsz * gsz, Kc)
    ref = torch.gather(pix, 2, keep[:, None, :, None].expand(B, T, n, Kc)).reshape(-1, Kc)
    assert torch.equal(cols[:, :Kc].float().cpu(), ref.bfloat16().float())
    assert float(cols[:, Kc:].float().abs().max()) == 0.0 if Kp > Kc else True
    # and the golden frames themselves through the kernel
    f = golden("transform")
    fr = torch.tensor(f["frames"]).unsqueeze(0)
    im = int(f["image"])
    keep1 = torch.arange(4, dtype=torch.int32).unsqueeze(0)
    c1 = torch.empty(3 * 4, 3 * 16 * 16, dtype=torch.bfloat16, device=DEV)
    K.patch_gather_u8(fr.to(DEV), keep1.to(DEV), c1, B=1, T=3, n=4, img=im, patch=16)
    out = torch.tensor(f["out"]).unsqueeze(0)  # [1, T, 3, 32, 32] from the reference
    px = out.reshape(1, 3, 3, 2, 16, 2, 16).permute(0, 1, 3, 5, 2, 4, 6).reshape(1, 3, 4, 768).reshape(-1, 768)
    assert torch.equal(c1.float().cpu(), px.bfloat16().float())


def test_step_on_uint8_frames_equals_step_on_fp32_clip(gpu):
    from tvts_amd import arch as A
    from tvts_amd.model._common import TVTSv2Base
    a = A.small_arch()
    oarch = O.tiny_arch(**a)
    P = O.synth_params(oarch, seed=2)
    m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
    m.load_state_dict(P, strict=True)
    b = O.synth_batch(oarch, B=3, T=2, seed=4, caption_len=9)
    g = torch.Generator().manual_seed(6)
    frames = torch.randint(0, 256, (3, 2, 76, 70, 3), generator=g, dtype=torch.uint8)
    crop = torch.tensor([[0, 3], [12, 6], [5, 0]])
    with torch.no_grad():
        te8, ve8, pr8 = m(dict(b, video=frames, crop=crop))
        te, ve, pr = m(dict(b, video=O.frames_to_video(frames, a["image"], crop)))
    assert torch.equal(ve8, ve) and torch.equal(te8, te) and torch.equal(pr8, pr)
    # centre crop when no offsets are given
    with torch.no_grad():
        _, vc8, _ = m(dict(b, video=frames))
        _, vc, _ = m(dict(b, video=O.frames_to_video(frames, a["image"])))
    assert torch.equal(vc8, vc) and not torch.equal(vc8, ve8)


@pytest.mark.parametrize("tag", ["wide", "tall", "same"])
def test_resize_fused_into_the_gather_bit_exact(gpu, golden, tag):
    """decoder-sized uint8 pictures -> Resize(48) (PIL nearest) -> CenterCrop(40) / a fixed RandomCrop offset -> ClipToTensor ->
    Normalize, all inside the tube-mask gather, against the reference's own chain (tests/golden/transform_resize.npz).  The odd
    size differences here (75 - 40) also pin CenterCrop's int(round(x / 2.)) = round-half-to-even offset."""
    from tvts_amd import hip as K
    from tvts_amd.data_loader.transforms import resize_tables
    f = golden("transform_resize")
    img, size = int(f["image"]), int(f["size"])
    frames = torch.tensor(f["frames_" + tag]).unsqueeze(0)  # [1, T, Hs, Ws, 3]
    T, hs, ws = frames.shape[1:4]
    tabs = resize_tables(hs, ws, size, DEV)
    assert [tabs[0].numel(), tabs[1].numel()] == list(f["resized_hw_" + tag])
    patch, g = 8, img // 8
    keep = torch.arange(g * g, dtype=torch.int32).unsqueeze(0)
    for crop, key in ((None, "out_" + tag), (torch.tensor([[3, 5]], dtype=torch.int32), "out_crop_" + tag)):
        cols = torch.empty(T * g * g, 3 * patch * patch, dtype=torch.bfloat16, device=DEV)
        K.patch_gather_u8(frames.to(DEV), keep.to(DEV), cols, B=1, T=T, n=g * g, img=img, patch=patch,
                          crop=None if crop is None else crop.to(DEV), resize=tabs)
        out = torch.tensor(f[key]).unsqueeze(0)  # [1, T, 3, img, img] from the reference
        px = out.reshape(1, T, 3, g, patch, g, patch).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, 3 * patch * patch)
        assert torch.equal(cols.float().cpu(), px.bfloat16().float()), key


def test_step_on_decoder_frames_with_resize(gpu):
    """model(data) with data['resize'] = 76: the step on raw 90 x 120 pictures equals the step on the clip the reference's
    Resize -> CenterCrop -> ToTensor -> Normalize would have produced (restated with the same index tables on the CPU)."""
    import numpy as np
    from tvts_amd import arch as A
    from tvts_amd.data_loader.transforms import pil_nearest_table, resize_sizes
    from tvts_amd.model._common import TVTSv2Base
    a = A.small_arch()
    oarch = O.tiny_arch(**a)
    m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
    m.load_state_dict(O.synth_params(oarch, seed=2), strict=True)
    b = O.synth_batch(oarch, B=3, T=2, seed=4, caption_len=9)
    frames = torch.randint(0, 256, (3, 2, 90, 121, 3), generator=torch.Generator().manual_seed(7), dtype=torch.uint8)
    size = 77
    h0, w0 = resize_sizes(90, 121, size)
    yt, xt = np.array(pil_nearest_table(90, h0)), np.array(pil_nearest_table(121, w0))
    resized = frames[:, :, yt][:, :, :, xt]
    with torch.no_grad():
        _, v8, _ = m(dict(b, video=frames, resize=size))
        _, vr, _ = m(dict(b, video=O.frames_to_video(resized, a["image"])))
    assert torch.equal(v8, vr)
