"""Host halves of the input pipeline against fixtures produced by the reference itself
(tests/golden/make_golden.py transform_resize / tokenize): Pillow's nearest-neighbour index tables + the reference's resize
size rule reproduce its Resize -> crop -> ClipToTensor -> Normalize chain bit for bit, and the caption cache returns the rows
clip.tokenize returns."""
import numpy as np
import pytest
import torch

from tvts_amd.data_loader.transforms import CaptionCache, pil_nearest_table, resize_sizes

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def chain(frames, size, img, top_left=None):
    """what the gather kernel computes, restated in numpy: resized(y, x) = frames[ytab[y], xtab[x]], crop, /255, normalise"""
    T, hs, ws, _ = frames.shape
    h0, w0 = resize_sizes(hs, ws, size)
    ytab, xtab = np.array(pil_nearest_table(hs, h0)), np.array(pil_nearest_table(ws, w0))
    y0, x0 = (int(round((h0 - img) / 2.)), int(round((w0 - img) / 2.))) if top_left is None else top_left  # CenterCrop :454-455
    px = frames[:, ytab[y0:y0 + img]][:, :, xtab[x0:x0 + img]].astype(np.float32)          # T, img, img, 3
    out = (px / np.float32(255.0) - np.array(MEAN, np.float32)) / np.array(STD, np.float32)
    return out.transpose(0, 3, 1, 2), (h0, w0)


@pytest.mark.parametrize("tag", ["wide", "tall", "same"])
def test_resize_crop_normalise_chain_bit_exact(golden, tag):
    f = golden("transform_resize")
    frames, img, size = f["frames_" + tag], int(f["image"]), int(f["size"])
    out, hw = chain(frames, size, img)
    assert list(hw) == list(f["resized_hw_" + tag])
    assert np.array_equal(out, f["out_" + tag])
    y1, x1 = [int(v) for v in f["crop_yx"]]
    out2, _ = chain(frames, size, img, (y1, x1))
    assert np.array_equal(out2, f["out_crop_" + tag])


def test_caption_cache_returns_the_tokenizer_rows(golden):
    f = golden("tokenize")
    caps = [str(c) for c in f["captions"]]
    table = {(c, True): torch.tensor(r) for c, r in zip(caps, f["tokens"])}
    table.update({(c, False): torch.tensor(r) for c, r in zip(caps[:5], f["tokens_short"])})
    calls = []

    def tokenizer(texts, truncate=False):  # stands in for clip.tokenize with the rows the real one produced
        calls.append(list(texts))
        return torch.stack([table[(t, truncate)] for t in texts])
    cc = CaptionCache(tokenizer)
    got = cc(caps, truncate=True)
    assert got.dtype == torch.int32 and torch.equal(got, torch.tensor(f["tokens"]))
    assert calls == [[caps[0], caps[1], caps[2], caps[4], caps[5]]]  # the repeated caption is tokenised once
    assert (got[:, 0] == 49406).all() and int(got[5].max()) == 49407 and int(got[5, -1]) == 49407  # truncated row ends in EOT
    again = cc(list(reversed(caps)), truncate=True)
    assert torch.equal(again, torch.tensor(f["tokens"]).flip(0)) and len(calls) == 1 and cc.hits == 7
    short = cc(caps[:5])  # another truncate flag is another key
    assert torch.equal(short, torch.tensor(f["tokens_short"])) and len(calls) == 2
    small = CaptionCache(tokenizer, max_entries=2)
    assert torch.equal(small(caps, truncate=True), torch.tensor(f["tokens"])) and len(small.rows) == 2
