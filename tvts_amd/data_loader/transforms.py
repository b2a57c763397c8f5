"""Host-side halves of the input pipeline that the device kernels complete.

* the Resize of the reference's transform chain -- ``video_transform.Resize(int(crop_size * 1.2))`` with its default
  interpolation 'nearest' on the PIL images ``TensorToNumpy`` hands over (v2/video_transforms/videoaug.py:12,21,
  video_transform.py:171-188,656-664, functional.py:47-64,71-78) -- as INDEX TABLES: Pillow's nearest-neighbour resize reads
  source column ``int(xo)`` for ``xo = 0.5 * s, += s`` (s = src / dst, accumulated in double; Pillow's ImagingScaleAffine).
  The tables are built here exactly that way (a few hundred integers per clip shape, cached) and the gather kernel
  (tvts_patch_gather_u8_resized) reads the decoder's uint8 frames through them: resize, crop, /255, normalise and the tube-mask
  gather are one pass over the kept patches only.
* ``CaptionCache``: the tokenised-caption cache keyed by caption text in front of ``clip.tokenize``
  (v2/CLIP/clip/clip.py:197-237): BPE-encoding a caption costs ~100 us of Python per call and the YT-Temporal transcripts
  repeat from epoch to epoch; rows are stored as int32 [context] and stacked clip-major as the trainer needs them.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Sequence, Tuple

import torch


def resize_sizes(im_h: int, im_w: int, size: int) -> Tuple[int, int]:
    """(new_h, new_w) of functional.get_resize_sizes (:71-78); a picture whose smaller side already equals `size` is left alone
    (resize_clip :49-53)."""
    if (im_w <= im_h and im_w == size) or (im_h <= im_w and im_h == size):
        return im_h, im_w
    if im_w < im_h:
        return int(size * im_h / im_w), size
    return size, int(size * im_w / im_h)


def pil_nearest_table(n_in: int, n_out: int) -> List[int]:
    """source index of every output index of PIL.Image.resize(..., NEAREST) along one axis"""
    s = n_in / n_out
    xo = s * 0.5
    tab = []
    for _ in range(n_out):
        tab.append(min(int(xo), n_in - 1))
        xo += s
    return tab


_TABLES: Dict[tuple, tuple] = {}


def resize_tables(hs: int, ws: int, size: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """device int32 (ytab [H0], xtab [W0]) for decoder pictures of hs x ws resized with video_transform.Resize(size)"""
    key = (hs, ws, size, str(device))
    t = _TABLES.get(key)
    if t is None:
        h0, w0 = resize_sizes(hs, ws, size)
        t = _TABLES[key] = (torch.tensor(pil_nearest_table(hs, h0), dtype=torch.int32, device=device),
                            torch.tensor(pil_nearest_table(ws, w0), dtype=torch.int32, device=device))
    return t


class CaptionCache:
    """tokenizer(texts, truncate=...) -> int32 [len(texts), context] with the rows of already-seen captions served from a bounded
    LRU map.  `tokenizer` is the reference's ``clip.tokenize`` (or anything with its signature); rows are bit-identical to
    calling it directly because each caption is tokenised on its own there too (clip.py:222-235)."""

    def __init__(self, tokenizer: Callable, max_entries: int = 1 << 20):
        self.tokenizer, self.max_entries = tokenizer, max_entries
        self.rows: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
        self.hits = self.misses = 0

    def __call__(self, texts: Sequence[str], truncate: bool = False, **kw) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        miss = [t for t in dict.fromkeys(texts) if (t, truncate) not in self.rows]
        if miss:
            toks = self.tokenizer(miss, truncate=truncate, **kw) if (truncate or kw) else self.tokenizer(miss)
            for t, row in zip(miss, toks):
                self.rows[(t, truncate)] = row.clone()
            while len(self.rows) > self.max_entries:
                self.rows.popitem(last=False)
        self.misses += len(miss)
        self.hits += len(texts) - len(miss)
        out = []
        for t in texts:
            r = self.rows.get((t, truncate))
            if r is None:  # evicted by this very call (cache smaller than the batch)
                r = (self.tokenizer([t], truncate=truncate, **kw) if (truncate or kw) else self.tokenizer([t]))[0]
            else:
                self.rows.move_to_end((t, truncate))
            out.append(r)
        return torch.stack(out)
