"""Step engine: explicit forward / backward of the TVTSv2 model over hand-written HIP kernels.

No autograd graph is built here: every activation that the backward pass needs is kept in a named
workspace buffer and the backward is written out by hand, kernel by kernel (tvts_amd.hip).  The
module classes in tvts_amd/model wrap this engine behind the reference's nn.Module surface.

Numerics: bf16 MFMA operands with fp32 accumulation; the residual stream, LayerNorm statistics,
embeddings, losses, gradients of parameters, master weights and optimizer state are fp32.

Reference being restated (file:line under v2/):
  forward wiring            model/model_dist_TVTSv2_ViT_B_16.py:61-116
  space-time ViT            model/video_encoder_ViT_B_16.py:94-124,176-235
  CLIP text tower           CLIP/clip/model.py:171-203,345-358
  transcript sorting head   model/sort_transformer.py:61-80,124-142
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import os

import torch

from . import hip as K
from .arch import is_mfma_weight, n_keep, param_shapes, patches_per_frame

CH = 1024  # flat-buffer alignment / optimizer chunk (elements)


class ParamStore:
    """All parameters in ONE flat fp32 buffer (+ flat grad, bf16 shadow, transposed bf16 shadow).

    nn.Parameters handed to the outside world are views into ``flat``; ``grad`` views are installed as
    ``.grad`` so an external optimizer (the reference entrypoint builds transformers.AdamW) sees them.
    """

    def __init__(self, arch: dict, device: torch.device):
        if arch.get("fp8_wgrad"):  # e4m3 weight gradients ride on the e4m3 forward and input-gradient operand copies
            arch["fp8"] = arch["fp8_dgrad"] = True
        self.arch = arch
        self.device = device
        self.shapes = param_shapes(arch)
        self.off: Dict[str, int] = {}
        off = 0
        for name, shape in self.shapes.items():
            self.off[name] = off
            off += -(-int(np.prod(shape)) // CH) * CH
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=device)
        # transposed copies of the MFMA weights
        self.toff: Dict[str, int] = {}
        toff, tiles = 0, []
        for name, shape in self.shapes.items():
            if not is_mfma_weight(name, shape):
                continue
            R, C = shape[0], int(np.prod(shape[1:]))
            self.toff[name] = toff
            for tr in range(-(-R // 64)):
                for tc in range(-(-C // 64)):
                    tiles.append((self.off[name], toff, R, C, tr, tc))
            toff += -(-R * C // CH) * CH
        self.shadow_t = torch.zeros(max(toff, 8), dtype=torch.bfloat16, device=device)
        rec = np.zeros(len(tiles), dtype=[("s", "<i8"), ("d", "<i8"), ("R", "<i4"), ("C", "<i4"), ("tr", "<i4"), ("tc", "<i4")])
        for i, t in enumerate(tiles):
            rec[i] = t
        self.n_tiles = len(tiles)
        self.tile_src = [t[0] for t in tiles]  # (ascending: parameter order) -> tiles of a flat range by bisection
        self.tile_table = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
        # patch-embedding GEMM: K = 3*p*p must be a multiple of 64 for the MFMA kernel; H/14 (588) gets a zero-padded copy
        self.conv_name = "video_model.patch_embed.proj.weight" if arch.get("family") == "v1" else "video_model.conv1.weight"
        kc = 3 * arch.get("tubelet", 1) * arch["patch"] ** 2
        self.conv_k = kc
        self.conv_kpad = -(-kc // 64) * 64
        self.conv_pad = (torch.zeros(arch["width"], self.conv_kpad, dtype=torch.bfloat16, device=device)
                         if self.conv_kpad != kc else None)
        self.w8: Dict[str, tuple] = {}
        self.w8t: Dict[str, tuple] = {}
        self._q8_table = None
        self.fp8_names = [n for n in self.shapes if n.startswith("video_model.transformer.resblocks.") and
                          n.endswith(("qkv.weight", "proj.weight", "c_fc.weight", "c_proj.weight"))]
        self.m: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self.shadow_version = -1
        self._views: Dict[str, torch.Tensor] = {}
        self._gviews: Dict[str, torch.Tensor] = {}

    def _n(self, name):
        return int(np.prod(self.shapes[name]))

    def p(self, name) -> torch.Tensor:
        v = self._views.get(name)
        if v is None:
            o = self.off[name]
            v = self._views[name] = self.flat[o:o + self._n(name)].view(self.shapes[name])
        return v

    def g(self, name) -> torch.Tensor:
        v = self._gviews.get(name)
        if v is None:
            o = self.off[name]
            v = self._gviews[name] = self.grad[o:o + self._n(name)].view(self.shapes[name])
        return v

    def g2d(self, name) -> torch.Tensor:
        s = self.shapes[name]
        return self.g(name).view(s[0], -1)

    def w(self, name) -> torch.Tensor:
        """bf16 shadow, 2-D [shape[0], rest] as stored."""
        o, s = self.off[name], self.shapes[name]
        return self.shadow[o:o + self._n(name)].view(s[0], -1)

    def wt(self, name) -> torch.Tensor:
        """bf16 shadow transposed, [rest, shape[0]]."""
        o, s = self.toff[name], self.shapes[name]
        return self.shadow_t[o:o + self._n(name)].view(-1, s[0])

    def refresh_transposed(self, s: int, e: int):
        """the transposed shadows of the MFMA weights inside the flat element range [s, e) (tiles are stored in parameter order)"""
        import bisect
        lo, hi = bisect.bisect_left(self.tile_src, s), bisect.bisect_left(self.tile_src, e)
        if hi > lo:
            K.transpose_batched(self.shadow, self.shadow_t, self.tile_table[lo * 32:hi * 32], hi - lo)

    def refresh_shadows(self, cast: bool = True, transposed: bool = True):
        """fp32 master -> bf16 shadow (+ transposed copies).  cast=False when the fused AdamW kernel
        already wrote the plain shadow; transposed=False when refresh_transposed has covered every range."""
        if cast:
            K.cast_f32_bf16(self.flat, self.shadow)
        if self.n_tiles and transposed:
            K.transpose_batched(self.shadow, self.shadow_t, self.tile_table, self.n_tiles)
        if self.conv_pad is not None:
            K.pad_rows_bf16(self.w(self.conv_name), self.conv_pad)
        if self.arch.get("fp8"):  # BASELINE config 4: e4m3 copies (+ per-tensor scales) of the ViT blocks' linear weights
            # ... and (fp8_dgrad) of their transposes, the input-gradient operands: the bf16 transposed shadow under the scale of the
            # master weight (a bf16 rounding past its amax saturates at +-448).  One table, three launches for all of them.
            if self._q8_table is None:
                dg = bool(self.arch.get("fp8_dgrad"))
                nw = len(self.fp8_names)
                scal = torch.zeros(3 * nw, dtype=torch.float32, device=self.device)  # amax | scale | scale_t per weight
                ents = []
                for i, name in enumerate(self.fp8_names):
                    s0 = self.shapes[name]
                    rows, cols = s0[0], int(np.prod(s0[1:]))
                    q = torch.empty(rows, cols, dtype=torch.uint8, device=self.device)
                    am, sc, sct = scal[3 * i:3 * i + 1], scal[3 * i + 1:3 * i + 2], scal[3 * i + 2:3 * i + 3]
                    self.w8[name] = (q, sc, am)
                    qt = None
                    if dg:
                        qt = torch.empty(cols, rows, dtype=torch.uint8, device=self.device)
                        self.w8t[name] = (qt, sct)
                    ents.append((self.p(name).view(rows, cols), q, self.wt(name) if dg else None, qt, am, sc, sct if dg else None))
                self._q8_table = K.quantize_fp8_multi_table(ents, self.device)
            K.quantize_fp8_multi(*self._q8_table)

    def w_conv(self) -> torch.Tensor:
        """bf16 patch-embedding weight [W, K padded to 64]."""
        return self.conv_pad if self.conv_pad is not None else self.w(self.conv_name)


_TEXT_NAMES = dict(ln1="ln_1", qkv_w="attn.in_proj_weight", qkv_b="attn.in_proj_bias", o_w="attn.out_proj.weight",
                   o_b="attn.out_proj.bias", ln2="ln_2", fc_w="mlp.c_fc.weight", fc_b="mlp.c_fc.bias",
                   pj_w="mlp.c_proj.weight", pj_b="mlp.c_proj.bias")
_SORT_NAMES = dict(ln1="norm1", qkv_w="attn.qkv.weight", qkv_b="attn.qkv.bias", o_w="attn.proj.weight",
                   o_b="attn.proj.bias", ln2="norm2", fc_w="mlp.fc1.weight", fc_b="mlp.fc1.bias",
                   pj_w="mlp.fc2.weight", pj_b="mlp.fc2.bias")


class Engine:
    def __init__(self, store: ParamStore):
        self.P = store
        self.arch = store.arch
        a = self.arch
        if a["tail"] not in ("all_tokens", "pooled_and_patches"):
            raise ValueError(f"unknown tail {a['tail']!r}")
        self.dh = a["width"] // a["heads"]            # ViT head dim: 64 (B/32, B/16) or 80 (H/14)
        self.dh_text = a["text_width"] // a["text_heads"]
        self.sort_width = a.get("sort_width", a["embed"])  # v2: the sort head works on the projected tokens (E); v1: on the ViT width
        self.dh_sort = self.sort_width // a["sort_heads"]
        for d, w, h in ((self.dh, a["width"], a["heads"]), (self.dh_text, a["text_width"], a["text_heads"]),
                        (self.dh_sort, self.sort_width, a["sort_heads"])):
            if d not in (64, 80) or d * h != w:
                raise NotImplementedError(f"attention kernels are built for head dim 64 and 80, not {w}/{h}")
        self.pooled_tail = a["tail"] == "pooled_and_patches"
        self.sort_used_rows_only = bool(a.get("sort_used_rows_only", True))  # last sort block on the rows the head reads (sort_forward)
        self.text_used_rows_only = bool(a.get("text_used_rows_only", a.get("sort_used_rows_only", True)))  # last text block: EOT rows
        self.has_sort_head = bool(a.get("sort_head", True))
        # Precision of the space-time blocks' residual stream (round 3).  Every consumer but the residual adds themselves reads
        # these tensors as bf16 anyway (LayerNorm outputs and GEMM operands are bf16): fp32 buys exact accumulation along the
        # 12 / 32 blocks, at 2.1 GB per block of HBM traffic in the forward + saved activations and 1.4 GB in the backward chain
        # (192 pairs).  Measured on B/16 against the fp32 oracle (experiments/dbg/grad_margins.py, profiles/r03_bf16_streams_ab.txt):
        #   gradient stream bf16 (OPT-IN since round 4, arch["bf16_grad_stream"], bench.py --bf16-grad-stream; round 3 had it on):
        #     +1.4 %; forward untouched; gradient norm -0.25 %, worst tensor cosine 0.9983 -- but the embedding-side gradients
        #     behind ln_pre's cancellation carry 2.4x the error of the fp32 chain (5.8 % rel-L2 instead of 2.4 %), and the golden
        #     gradient slices needed their gate widened from 0.08 to 0.12 to pass: parity is the first gate, 1 % is not worth it;
        #   residual stream bf16 as well (OPT-IN, arch["bf16_residual"], bench.py --bf16-residual): +5 % in total (1354-1372
        #     pairs/s); video embedding rel-L2 0.89 % (gate 2 %; 0.39 % in fp32), row cosine 0.99995 (gate 0.9995), but the
        #     |d loss| <= 1e-2 gate fails on one small 3-pair NT = 1 configuration (0.0155) -- parity is the first gate, so it
        #     is not the default.
        # Round 5, the HYBRID stream (arch["hybrid_stream"], the default for v2): both streams bf16, EXCEPT the CLS token's row of
        # every clip, which is carried in fp32 in compact [B, W] side arrays.  experiments/dbg/bf16_residual_rows.py (the fp32 oracle
        # with bf16 roundings at the engine's stream points) shows where the error of a bf16 stream comes from: rounding the 784
        # patch rows of a clip costs |d loss1| 2e-4 ... 1.7e-3 (2 ... 12 blocks), rounding the ONE CLS row 3.4e-3 ... 5e-3 -- the video
        # embedding is read from that row, and its error reaches every other token through the attention, while the patch rows'
        # independent errors average out in it; for the gradient stream likewise (positional-embedding gradient rel-L2 1e-4 ... 1.3e-3
        # with the CLS row exact against 2 ... 3.7e-2 with every row rounded).  So 784 of 785 rows take the bf16 bytes and the loss /
        # gradient errors stay at the fp32 streams' level: the forward CLS rows through two [B, K] GEMMs per block with fp32 residual
        # (`_cls_lin`), the LayerNorms read / write them beside the stream (tvts_layernorm_{fwd,bwd}_cls).
        hybrid = a.get("hybrid_stream", None)
        if hybrid is None and os.environ.get("TVTS_HYBRID_STREAM"):
            hybrid = os.environ["TVTS_HYBRID_STREAM"] != "0"
        if hybrid is None:  # default: on, unless one of the older stream options is asked for explicitly
            hybrid = a.get("family") != "v1" and "bf16_residual" not in a and "bf16_grad_stream" not in a
        self.cls32 = bool(hybrid) and a.get("family") != "v1"
        self.bf16_residual = (bool(a.get("bf16_residual", False)) or self.cls32) and a.get("family") != "v1"
        self.bf16_grad_stream = bool(a.get("bf16_grad_stream", False)) or self.bf16_residual
        # Weight gradients of the space-time blocks on a SIDE STREAM (round 4).  dW = dY^T X depends on dY only, not on the
        # input-gradient chain that continues from dY, so the weight-gradient kernels can run beside the chain's NT GEMMs,
        # attention and LayerNorm backwards.  At the reference's own per-GPU batches (12 / 24 pairs: 111 ... 444 output tiles
        # for 256 persistent blocks) the chain leaves a third to a half of the CUs idle; the side stream's kernels fill them.
        # MEASURED (profiles/r04_wgrad_side_stream.txt) and NOT the default: the kernels do overlap in the replayed graph, but a
        # weight-gradient block (64 KiB LDS, two per CU) and an NT block (160 KiB, the whole CU) cannot share a CU, so the pairs
        # mostly take turns (NT 63 us + TN 50 us alone -> 112 us side by side), and every fork / join edge between the two
        # streams costs the graph 5-15 us: 12 / 24 / 48 / 192 pairs run 667.9 / 933.4 / 1130 / 1325 pairs/s with it against
        # 692.9 / 958.5 / 1154 / 1351 without.  arch["wgrad_stream"] = True switches it on (bit-identical gradients,
        # tests/test_bench_path_gpu.py::test_wgrad_side_stream_gives_the_same_bits).
        self.wgrad_stream = a.get("wgrad_stream", False)
        # The MLPs' saved tensor is act'(pre-activation), not the pre-activation (round 5, TVTS_GEMM_SIDE_DERIV; arch["gate_deriv"] =
        # False stores the pre-activation as rounds 1-4 did): the forward epilogue evaluates the sigmoid / erf parts for the
        # activation anyway and adds two multiply-adds for the derivative; the input gradient's gate epilogue then multiplies by the
        # stored bf16 value instead of evaluating the transcendentals again.  Nothing else reads the tensor.
        self.gate_deriv = bool(a.get("gate_deriv", os.environ.get("TVTS_GATE_DERIV", "1") != "0"))
        # The text tower on its own stream beside the ViT (round 5).  The two towers share nothing until the loss (the sort head reads
        # DETACHED caption embeddings, model_dist_TVTSv2_ViT_B_16.py:61-95): the text tower's forward (12 blocks of 5-25 us kernels
        # on <= 48 tiles) and its three trainable blocks' backward are latency-bound chains -- in the replayed 12-pair step every one
        # of their ~150 kernels waits 10-20 us for its predecessor's completion signal (profiles/r04_timeline_b12.txt) -- that fit
        # beside the ViT's kernels: ONE fork and ONE join per direction, scratch buffers of their own (hip.lane).
        # arch["text_side"]: True / False, None = on
        ts = a.get("text_side", None)
        if ts is None and os.environ.get("TVTS_TEXT_SIDE"):  # (A/B runs of whole test files / tools without touching their arch dicts)
            ts = os.environ["TVTS_TEXT_SIDE"] != "0"
        self.text_side = (True if ts is None else bool(ts)) and a.get("family") != "v1"
        self._txt_stream = None
        # The six weight gradients of a ViT block in ONE grouped launch + one reduce launch (round 4, tvts_gemm_tn_bf16_grouped):
        # arch["tn_grouped"] True / False, None = automatic (up to TN_GROUPED_MAX_ROWS token rows per GPU, where a single weight
        # gradient fills a fraction of a round of the chip and its launch ramp is a third of its time)
        self.tn_grouped = a.get("tn_grouped", None)
        self._tn_groups: Dict[object, object] = {}
        self._tn_group_ws = None
        # e4m3 weight gradients (BASELINE config 5): per-tensor delayed scaling of every e4m3 operand copy, see _q8 below
        self.fp8_wgrad = bool(a.get("fp8_wgrad", False))
        self.fp8_q8_only = self.fp8_wgrad and bool(a.get("fp8_q8_only", True))
        self._x8_only = set()
        self._f8_ids: Dict[str, int] = {}
        self._f8_scale = self._f8_amax = None
        self._f8_tensor_mode = False
        self._x8: Dict[str, tuple] = {}
        self._x8_ready: Dict[str, tuple] = {}   # operand copies written by a producer's epilogue, waiting for their consumer
        self._dy8_ready: Dict[str, tuple] = {}
        self._wg_stream = None
        self._wg_ws = None
        self.dev = store.device
        if self.dev.type == "cuda":
            K.warm_scratch(self.dev)
        self.buf: Dict[str, torch.Tensor] = {}
        self._back: Dict[str, torch.Tensor] = {}  # the allocations behind buf (grow-only, see _b)
        self._seen: Dict[str, int] = {}           # name -> the forward() count at which its shape last changed
        self._tick = 0
        self.requires_grad = {name: True for name in store.shapes}
        self.ctx: dict = {}
        self.grad_ready = None  # optional callback(start, end): flat grad range is final (GradSync.reduce_range); may return a wait()
        self.param_ready = None  # optional callback(start, end, wait): the range may be stepped (FusedHFAdamW.step_range)
        self.embeds_ready = None  # optional callback(text_emb, video_emb): both embeddings exist, the sort head has not run yet
        self._ranges: dict = {}

    def _ready(self, *prefixes: str):
        """The gradients of every parameter whose name starts with one of `prefixes` are final: hand their flat ranges
        (contiguous in state-dict order, whichever registration order the architecture uses) to the gradient sync --
        maximal runs of TRAINABLE tensors only: the frozen text layers below the tune range (train_dist..:89-96, ~28 M
        parameters of all-zero gradient for ViT-B/16) never travel."""
        if self.grad_ready is None and self.param_ready is None:
            return
        for lo, hi in self.trainable_runs(prefixes):
            wait = self.grad_ready(lo, hi) if self.grad_ready is not None else None
            if self.param_ready is not None:  # the optimizer's update of the range, beside the remaining backward
                self.param_ready(lo, hi, wait)

    def trainable_runs(self, prefixes):
        P = self.P
        ent = self._ranges.get(prefixes)
        if ent is None:
            names = [n for n in P.shapes if n.startswith(prefixes)]
            lo = min(P.off[n] for n in names)
            hi = max(P.off[n] + -(-P._n(n) // CH) * CH for n in names)
            inside = [n for n in P.shapes if lo <= P.off[n] < hi]
            assert inside == names, f"parameters {prefixes} are not contiguous in the flat buffer"
            ent = self._ranges[prefixes] = dict(names=names, sig=None, runs=None)
        sig = tuple(self.requires_grad[n] for n in ent["names"])
        if sig != ent["sig"]:
            runs = []
            for n, tr in zip(ent["names"], sig):
                if not tr:
                    continue
                s, e = P.off[n], P.off[n] + -(-P._n(n) // CH) * CH
                if runs and runs[-1][1] == s:
                    runs[-1][1] = e
                else:
                    runs.append([s, e])
            ent["sig"], ent["runs"] = sig, [tuple(r) for r in runs]
        return ent["runs"]

    # ------------------------------------------------------------------ workspace
    def _b(self, name, shape, dtype=torch.bfloat16, zero=False):
        """A named workspace tensor.  Its storage only ever GROWS: a smaller shape is a view of the first elements of the same
        allocation, so a loop that alternates batches of two sizes (the reference's YT / WebVid loaders, trainer.py:463) keeps
        every buffer's address -- no allocator traffic after the first pass over both, and plans keyed by buffer addresses (the
        grouped weight gradients) stay valid."""
        t = self.buf.get(name)
        shape = tuple(int(s) for s in shape)
        if t is not None and tuple(t.shape) == shape and t.dtype == dtype:
            self._seen[name] = self._tick  # (live in this step: a later request for ANOTHER shape under this name must not alias it)
            if zero:
                K.zero_(t)
            return t
        n = 1
        for s in shape:
            n *= s
        seen, self._seen[name] = self._seen.get(name), self._tick
        if seen == self._tick:
            # the name changes shape WITHIN one forward / backward pair: both tensors may be live, so they must not share memory
            t = self.buf[name] = torch.empty(shape, dtype=dtype, device=self.dev)
        else:
            back = self._back.get(name)
            if back is None or back.dtype != dtype or back.numel() < n:
                back = self._back[name] = torch.empty(max(n, 1), dtype=dtype, device=self.dev)
            t = self.buf[name] = back[:n].view(shape)
        if zero:
            K.zero_(t)
        return t

    def _f(self, name, shape, zero=False):
        return self._b(name, shape, torch.float32, zero)

    def _ones(self, n):
        """a vector of n ones (the bias gradients as 1^T dy products): a constant made once per length, never written again"""
        t = self.buf.get(("ones", n))
        if t is None:
            t = self.buf[("ones", n)] = torch.ones(n, dtype=torch.float32, device=self.dev)
        return t

    # ------------------------------------------------------------------ e4m3 copies (BASELINE config 4 / 5)
    # Two regimes of the e4m3 operand copies:
    #   per TOKEN  (arch["fp8"], arch["fp8_dgrad"]): one scale per row, written by the producing LayerNorm or by a one-pass row
    #              quantiser, applied per row in the GEMM epilogue; the weight gradients stay bf16;
    #   per TENSOR (arch["fp8_wgrad"], round 4): the weight gradient contracts over the tokens, so its operands can only carry one
    #              scale per tensor -- and a copy under one scale serves the forward / input-gradient GEMMs as well.  Delayed
    #              scaling: every producer quantises under the tensor's scale of LAST step (amax / 448) and records this step's
    #              amax; end_step() turns the maxima into the next scales.  The first step has no scales yet: it runs per token
    #              with bf16 weight gradients while the maxima are recorded ("calibration"), and end_step() switches over.
    def _f8_slot(self, name):
        """(scale, amax) device scalars of tensor `name` (views of two flat arrays, so that one kernel updates all of them)"""
        i = self._f8_ids.get(name)
        if i is None:
            if self._f8_scale is None:
                cap = 16 * (self.arch["layers"] + 2)
                self._f8_scale = torch.zeros(cap, dtype=torch.float32, device=self.dev)
                self._f8_amax = torch.zeros(cap, dtype=torch.float32, device=self.dev)
            i = self._f8_ids[name] = len(self._f8_ids)
            assert i < self._f8_scale.numel(), "more e4m3 tensors than the scale table holds"
        return self._f8_scale[i:i + 1], self._f8_amax[i:i + 1]

    def end_step(self):
        """Called once per optimisation step (StepRunner.run) after the backward: this step's amax values become the per-tensor
        scales of the next step, and from the second step on the e4m3 copies are per-tensor and the weight gradients e4m3."""
        if self.fp8_wgrad and self._f8_scale is not None:
            K.fp8_update_scales(self._f8_amax, self._f8_scale)
            self._f8_tensor_mode = True

    def _q8(self, M, W, name, persistent=False):
        """Buffers for the e4m3 copy of tensor `name` [M, W]: (bytes, scale operand for the GEMMs, kwargs for the producing kernel).
        persistent: the bytes outlive the next producer of the same width (an activation the weight gradient reads in the backward)."""
        q = self._b(("fp8.x." + name) if (persistent and self.fp8_wgrad) else "fp8.q%d" % W, (M, W), torch.uint8)
        if not self.fp8_wgrad:
            rs = self._f("fp8.row_scale", (max(M, 2),))  # never a 1-element tensor: that means "one scale for the tensor"
            return q, rs, dict(row_scale=rs)
        sc, am = self._f8_slot(name)
        if self._f8_tensor_mode:
            return q, sc, dict(tscale=sc, amax=am)
        rs = self._f("fp8.row_scale", (max(M, 2),))
        return q, rs, dict(row_scale=rs, amax=am)

    # ------------------------------------------------------------------ linear helpers
    def _wgrad_side(self, M) -> bool:
        return bool(self.wgrad_stream)

    def _side(self):
        """(side stream, its own split-partial workspace): the weight-gradient kernels of one stream share one workspace, the two
        streams never do"""
        if self._wg_stream is None:
            self._wg_stream = torch.cuda.Stream(device=self.dev)
            self._wg_ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=self.dev)
        return self._wg_stream, self._wg_ws

    def _q8_out(self, M, W, name, persistent=False):
        """kwargs that make an e4m3 GEMM's epilogue write the per-tensor e4m3 copy of its bf16 result (tensor mode only), and the
        (bytes, scale) pair its consumers take"""
        # OPT-IN (arch["fp8_epilogue_copies"]): measured on H/14, 16 frames, 48 pairs (profiles/r04_kernel_summary_h14_b48_fp8_wgrad.txt):
        # the two quantiser passes it removes are worth 8.6 ms per step, the two epilogues that take their work over get 105 / 120 us
        # slower per launch (7.2 ms: they are the epilogue-bound GELU / gate forms, and the extra stores spill 30-40 bytes more) --
        # 295.3 against 295.8 ms per step, not worth a second path by default
        # Round 5, arch["fp8_q8_only"] (default with fp8_wgrad): the epilogue writes the e4m3 copy INSTEAD of the bf16 result -- in the
        # per-tensor regime every consumer of the MLP's GELU output (fc2 forward, fc2 weight gradient) and of the gated hidden gradient
        # (fc1 input and weight gradient, bias gradient on the matrix pipe) multiplies the e4m3 bytes: one byte per element leaves
        # the epilogue where round 4 wrote two and a quantiser pass read them again
        if not (self.fp8_wgrad and self._f8_tensor_mode and (self.arch.get("fp8_epilogue_copies") or self.fp8_q8_only)):
            return {}, None
        q, sc, kw = self._q8(M, W, name, persistent=persistent)
        return dict(q8out=q, q8_scale=sc, q8_amax=kw["amax"]), (q, sc)

    def _lin(self, a, wname, bname, out, M, a8=None, q8_for=None, **epi):
        """q8_for: the weight whose GEMMs read `out` as their activation; in the per-tensor e4m3 regime the epilogue of this GEMM
        then writes that operand copy itself (the MLP's GELU output), instead of a quantiser pass over `out`."""
        if epi.get("preact") is not None and self.gate_deriv:
            epi["side_deriv"] = True  # the side output holds act'(x): the gate of the backward multiplies by it as is
        if wname in self.P.w8:  # fp8 weight/activation path (BASELINE config 4): forward GEMMs of the ViT blocks
            w8, ws, _ = self.P.w8[wname]
            if a8 is None:
                a8 = self._x8_ready.pop(wname, None)
            if q8_for is not None and q8_for in self.P.w8:
                kw8, nxt = self._q8_out(M, out.shape[1], "x." + q8_for, persistent=True)
                epi = dict(epi, **kw8)
                if nxt is not None:
                    self._x8_ready[q8_for] = nxt
                    if self.fp8_q8_only and self.requires_grad[q8_for]:
                        epi["store_out"] = False       # `out` (bf16) is not written: its consumers read nxt
                        self._x8_only.add(q8_for)
            if a8 is None:  # activations that do not come out of a LayerNorm: one pass
                q, sa, kw = self._q8(M, a.shape[1], "x." + wname, persistent=True)
                K.quantize_fp8_rows(a[:M], q=q, **kw)
                a8 = (q, sa)
            self._x8[wname] = a8
            K.gemm_nt_fp8(a8[0], a8[1], w8, ws, out[:M], bias=self.P.p(bname) if bname else None, **epi)
            return
        K.gemm_nt(a, self.P.w(wname), out, M=M, bias=self.P.p(bname) if bname else None, **epi)

    def _cls_lin(self, a, S, wname, bname, res_c, out_c):
        """The CLS rows of the hybrid stream through a residual-adding linear layer: out_c [B, N] = res_c + a[b * S, :] @ W^T + bias,
        fp32 result and fp32 residual (the stream's own rows take the bf16 residual epilogue).  a: the [B * S, K] bf16 operand of the
        block's GEMM; its CLS rows are addressed as a [B, K] matrix with the row stride S * K -- the same operand bytes and the same
        bf16 weight shadow, always on the bf16 MFMA path (also when the block's GEMMs multiply e4m3 copies)."""
        Bc = out_c.shape[0]
        if wname in self._x8_only:  # the operand only exists as e4m3 bytes (arch["fp8_q8_only"]): the same rows of that copy
            q, sa = self._x8[wname]
            w8, ws, _ = self.P.w8[wname]
            K.gemm_nt_fp8(q[:Bc * S].view(Bc, -1)[:, :q.shape[1]], sa, w8, ws, out_c, bias=self.P.p(bname) if bname else None, residual=res_c)
            return
        a_c = a[:Bc * S].view(Bc, -1)[:, :a.shape[1]]
        K.rows_linear(a_c, self.P.w(wname), out_c, bias=self.P.p(bname) if bname else None, residual=res_c)

    # measured (profiles/r04_tn_grouped.txt): B/16 +1.5 % at 12 pairs (M = 9 420), -0.4 % at 24, -2 % at 48; B/32 +1.2 % at its 24 pairs
    # (M = 9 432); H/14 at its 2 pairs (M = 2 434, 1 600 tiles of 128 x 128 per block of the model) -9 %: the window is where it pays
    TN_GROUPED_MIN_ROWS, TN_GROUPED_MAX_ROWS = 6000, 12000

    def _tn_grouped_on(self, M) -> bool:
        on = self.tn_grouped
        if on is None:
            on = self.TN_GROUPED_MIN_ROWS <= M <= self.TN_GROUPED_MAX_ROWS
        return bool(on) and not self.fp8_wgrad and not self.wgrad_stream

    def _tn_group_run(self, key, problems):
        """launch the deferred weight gradients of one block together.  Plans are cached per (block, problem signature): the
        reference's loop alternates two loaders with different clip lengths (trainer.py:463), so a block sees two signatures in
        turn -- replacing the plan would rebuild and re-upload it twelve times per step; a bounded number is kept per block."""
        if not problems:
            return
        sig = tuple((pr["p"].data_ptr(), pr["q"].data_ptr(), pr["out"].data_ptr(), pr["M"],
                     pr["colsum"].data_ptr() if pr["colsum"] is not None else 0) for pr in problems)
        plans = self._tn_groups.setdefault(key, {})
        grp = plans.get(sig)
        if grp is None:
            if self._tn_group_ws is None:
                self._tn_group_ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=self.dev)
            if len(plans) >= self.TN_GROUP_PLANS_PER_BLOCK:
                # the oldest plan no hipGraph refers to (a last partial batch, a loader that is gone); a captured plan's device table
                # and staging memory are addresses inside the graph, so it stays
                old = next((k for k, g in plans.items() if not g.captured), None)
                if old is not None:
                    plans.pop(old)
            grp = plans[sig] = K.TnGroup(problems, self._tn_group_ws, splits=int(self.arch.get("tn_group_splits", 0)))
        grp.run()

    TN_GROUP_PLANS_PER_BLOCK = 4

    def _lin_bwd(self, dy, a_in, wname, bname, d_in, M, dy8=None, side=False, q8_for=None, defer=None, **epi):
        """dW += dy^T a_in, db += colsum(dy) (if trainable); d_in = dy W (optional, with epilogue).  dy8: the e4m3 copy of dy
        when its producer already wrote one (_ln_bwd with fp8_for).  side: the weight gradient is launched on the side stream
        behind everything the current stream has queued so far (dy is complete there); the caller keeps dy and a_in untouched
        until it has joined the side stream."""
        want_b = bool(bname) and self.requires_grad[bname]
        if epi.get("gate_h") is not None and epi.get("gate_act") not in (None, "add") and self.gate_deriv:
            epi["side_deriv"] = True  # gate_h is the derivative the forward's epilogue stored (see _lin)
        f8w = self.fp8_wgrad and self._f8_tensor_mode and wname in self.P.w8t and wname in self._x8
        if dy8 is None:
            dy8 = self._dy8_ready.pop(wname, None)
        if (f8w or (d_in is not None and wname in self.P.w8t)) and dy8 is None:
            q, sa, kw = self._q8(M, dy.shape[1], "dy." + wname)
            K.quantize_fp8_rows(dy[:M], q=q, **kw)
            dy8 = (q, sa)
        if self.requires_grad[wname]:  # bias gradient (column sums of dy) rides along in the same kernel
            if defer is not None and not f8w:  # launched with the block's other weight gradients (the caller keeps dy untouched till then)
                defer.append(dict(p=dy, q=a_in, out=self.P.g2d(wname), M=M, accumulate=True, colsum=self.P.g(bname) if want_b else None))
            elif f8w:  # e4m3 weight gradient: both operands under one scale per tensor, the bias gradient from the same bytes
                x8 = self._x8[wname]
                K.gemm_tn_fp8(dy8[0], dy8[1], x8[0], x8[1], self.P.g2d(wname), M=M, accumulate=True,
                              colsum=self.P.g(bname) if want_b else None)
            elif side:
                ws, scratch = self._side()
                ws.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(ws):
                    K.gemm_tn(dy, a_in, self.P.g2d(wname), M=M, accumulate=True, colsum=self.P.g(bname) if want_b else None,
                              workspace=scratch)
            else:
                K.gemm_tn(dy, a_in, self.P.g2d(wname), M=M, accumulate=True, colsum=self.P.g(bname) if want_b else None)
        elif want_b:
            K.colsum(dy, self.P.g(bname), M=M)
        if d_in is not None:
            if wname in self.P.w8t:  # e4m3 input gradient (arch["fp8_dgrad"]): the transposed weight's e4m3 copy
                w8t, wst = self.P.w8t[wname]
                if q8_for is not None and q8_for in self.P.w8t and epi.get("gate_h") is not None:
                    # the gated input gradient is the output gradient of layer q8_for: its per-tensor e4m3 copy leaves this epilogue
                    kw8, nxt = self._q8_out(M, d_in.shape[1], "dy." + q8_for)
                    epi = dict(epi, **kw8)
                    if nxt is not None:
                        self._dy8_ready[q8_for] = nxt
                        if self.fp8_q8_only and self.requires_grad[q8_for] and q8_for in self._x8:
                            epi["store_out"] = False   # the gated gradient exists as e4m3 bytes only
                K.gemm_nt_fp8(dy8[0], dy8[1], w8t, wst, d_in[:M], **epi)
                return
            K.gemm_nt(dy, self.P.wt(wname), d_in, M=M, **epi)

    def _ln(self, x, name, eps, y, tag, rows=None, M=None, fp8_for=None, cls_x=None):
        """fp8_for: name of the weight the output feeds; when that GEMM runs in fp8 the LayerNorm also emits the e4m3 copy
        (returned, to be handed to _lin as a8).  cls_x: the fp32 CLS rows of the hybrid stream x (read instead of x's own CLS
        rows, which receive their bf16 rounding)."""
        M = (rows.numel() if rows is not None else x.shape[0]) if M is None else M
        if cls_x is not None:
            kw_cls = dict(cls_x=cls_x, cls_period=M // cls_x.shape[0])
        else:
            kw_cls = {}
        mean, rstd = self._f(tag + ".mean", (M,)), self._f(tag + ".rstd", (M,))
        a8, kw = None, {}
        if fp8_for is not None and fp8_for in self.P.w8:
            q, sa, kw = self._q8(M, y.shape[1], "x." + fp8_for, persistent=True)
            a8, kw = (q, sa), dict(kw, q8=q)
            # arch["fp8_q8_only"]: under per-tensor scales the forward GEMM and the weight gradient both multiply the e4m3 bytes,
            # the bf16 output would be written and never read (8-column form of the kernel: bf16 rows, W % 8 == 0)
            if (self.fp8_q8_only and self.fp8_wgrad and self._f8_tensor_mode and fp8_for in self.P.w8t and rows is None
                    and x.dtype == torch.bfloat16 and y.shape[1] % 8 == 0 and y.shape[1] <= 1536 and q.stride(0) % 8 == 0):
                y = None
        K.layernorm_fwd(x, self.P.p(name + ".weight"), self.P.p(name + ".bias"), eps, y, mean, rstd, rows=rows, M=M, **kw, **kw_cls)
        return a8

    def _ln_bwd(self, dy, x, name, tag, dx, dx_bf16=None, res1=None, res2=None, rows=None, M=None, fp8_for=None,
                cls_x=None, cls_res1=None, cls_dx=None):
        """fp8_for: name of the weight whose input-gradient GEMM consumes dx_bf16; when that GEMM runs on e4m3 operands the
        LayerNorm backward also emits the e4m3 copy (returned, to be handed to _lin_bwd as dy8).  cls_*: the fp32 CLS rows of the
        hybrid stream (input, incoming stream gradient, outgoing stream gradient)."""
        tr = self.requires_grad[name + ".weight"]
        d8, kw = None, {}
        if cls_x is not None or cls_res1 is not None or cls_dx is not None:
            nc = (cls_x if cls_x is not None else cls_dx if cls_dx is not None else cls_res1).shape[0]
            kw = dict(cls_period=(x.shape[0] if M is None else M) // nc, cls_x=cls_x, cls_res1=cls_res1, cls_dx=cls_dx)
        if (fp8_for is not None and fp8_for in self.P.w8t and rows is None and dx_bf16 is not None and dy.dtype == torch.bfloat16
                and not (res2 is not None and res1 is None)):
            q, sa, kw8 = self._q8(dx_bf16.shape[0] if M is None else M, dx_bf16.shape[1], "dy." + fp8_for)
            d8, kw = (q, sa), dict(kw, q8=q, **kw8)
        K.layernorm_bwd(dy, x, self.buf[tag + ".mean"], self.buf[tag + ".rstd"], self.P.p(name + ".weight"), dx,
                        dx_bf16=dx_bf16, res1=res1, res2=res2, dgamma=self.P.g(name + ".weight") if tr else None,
                        dbeta=self.P.g(name + ".bias") if tr else None, rows=rows, M=M, **kw)
        return d8

    # ------------------------------------------------------------------ generic pre-LN block (text tower, sort head)
    def _block_fwd(self, pre, nm, x_in, x_out, tag, M, Wd, heads, Bn, S, causal, act, eps):
        hd = Wd // heads
        ln1 = self._b(tag + ".ln1", (M, Wd))
        self._ln(x_in, pre + nm["ln1"], eps, ln1, tag + ".ln1")
        qkv = self._b(tag + ".qkv", (M, 3 * Wd))
        self._lin(ln1, pre + nm["qkv_w"], pre + nm["qkv_b"], qkv, M)
        att = self._b(tag + ".att", (M, Wd))
        lse = self._f(tag + ".lse", (M, heads))
        K.attn_fwd("full", qkv, att, lse, B=Bn, heads=heads, S=S, causal=causal, head_dim=hd)
        mid = self._f(tag + ".mid", (M, Wd))
        self._lin(att, pre + nm["o_w"], pre + nm["o_b"], mid, M, residual=x_in)
        ln2 = self._b(tag + ".ln2", (M, Wd))
        self._ln(mid, pre + nm["ln2"], eps, ln2, tag + ".ln2")
        h = self._b(tag + ".h", (M, 4 * Wd))
        a = self._b(tag + ".a", (M, 4 * Wd))
        self._lin(ln2, pre + nm["fc_w"], pre + nm["fc_b"], a, M, act=act, preact=h)
        self._lin(a, pre + nm["pj_w"], pre + nm["pj_b"], x_out, M, residual=mid)

    def _block_bwd(self, pre, nm, x_in, dx, dxb, dx_in, dxb_in, tag, M, Wd, heads, Bn, S, causal, act, scr):
        """dx/dxb: grad wrt block output (fp32 / bf16).  Writes grad wrt block input to dx_in/dxb_in."""
        B_ = self.buf
        hd = Wd // heads
        dh = self._b(scr + ".dh", (M, 4 * Wd))
        dln = self._b(scr + ".dln", (M, Wd))
        self._lin_bwd(dxb, B_[tag + ".a"], pre + nm["pj_w"], pre + nm["pj_b"], dh, M, gate_h=B_[tag + ".h"], gate_act=act)
        self._lin_bwd(dh, B_[tag + ".ln2"], pre + nm["fc_w"], pre + nm["fc_b"], dln, M)
        dmid = self._f(scr + ".dmid", (M, Wd))
        dmidb = self._b(scr + ".dmidb", (M, Wd))
        self._ln_bwd(dln, B_[tag + ".mid"], pre + nm["ln2"], tag + ".ln2", dmid, dx_bf16=dmidb, res1=dx)
        datt = self._b(scr + ".datt", (M, Wd))
        self._lin_bwd(dmidb, B_[tag + ".att"], pre + nm["o_w"], pre + nm["o_b"], datt, M)
        dqkv = self._b(scr + ".dqkv", (M, 3 * Wd))
        delta = self._f(scr + ".delta", (M, heads))
        qkv, lse = B_[tag + ".qkv"], B_[tag + ".lse"]
        K.attn_bwd("full", qkv, datt, B_[tag + ".att"], lse, delta, dqkv, B=Bn, heads=heads, S=S, causal=causal, head_dim=hd)
        self._lin_bwd(dqkv, B_[tag + ".ln1"], pre + nm["qkv_w"], pre + nm["qkv_b"], dln, M)
        self._ln_bwd(dln, x_in, pre + nm["ln1"], tag + ".ln1", dx_in, dx_bf16=dxb_in, res1=dmid)

    # ------------------------------------------------------------------ text tower
    @staticmethod
    def eot_index(eot_rows, L):
        """(None, int32 position of the EOT token inside its caption).  prepare_batch hands the positions over with the batch (made on
        the host, so that their lifetime is the batch's -- and that of a hipGraph captured over it); this form derives them from
        the rows for callers that drive text_forward by hand."""
        pos = (eot_rows - torch.arange(eot_rows.numel(), device=eot_rows.device, dtype=torch.int32) * L).contiguous()
        return None, pos

    def text_forward(self, ids_dev, eot_rows, N, L, eot_index=None):
        """eot_index: Engine.eot_index(eot_rows, L) kept by the caller (prepare_batch); made here when absent (eager use only)."""
        a = self.arch
        Wt, M = a["text_width"], N * L
        x = self._f("txt.x0", (M, Wt))
        K.text_embed(ids_dev, self.P.p("text_token_embedding.weight"), self.P.p("text_positional_embedding"), x, N=N, L=L)
        lnf = self._f("txt.lnf", (N, Wt))
        last = a["text_layers"] - 1
        for l in range(a["text_layers"]):
            if l == last and self.text_used_rows_only:
                # the model reads the last block's output at the EOT token of every caption only (CLIP/clip/model.py:343-354)
                ht, hd = a["text_heads"], Wt // a["text_heads"]
                rows64, pos = self._text_eot = eot_index if eot_index is not None else self.eot_index(eot_rows, L)
                xr = self._used_rows_fwd(f"text_model.resblocks.{l}.", _TEXT_NAMES, x, f"txt{l}", M, Wt, ht, eot_rows, a["act"], 1e-5,
                                         lambda qkv, att, lse: K.attn_fwd_rowq(qkv, pos, att, lse, B=N, heads=ht, S=L, head_dim=hd))
                self._ln(xr, "text_ln_final", 1e-5, lnf, "txt.lnf")
                break
            xo = self._f(f"txt.x{l + 1}", (M, Wt))
            self._block_fwd(f"text_model.resblocks.{l}.", _TEXT_NAMES, x, xo, f"txt{l}", M, Wt, a["text_heads"], N, L, True,
                            a["act"], 1e-5)
            x = xo
        else:
            self._ln(x, "text_ln_final", 1e-5, lnf, "txt.lnf", rows=eot_rows)
        t = self._f("txt.t", (N, a["embed"]))
        K.gemm_small(lnf, self.P.p("text_projection"), t, M=N, N=a["embed"], K=Wt, sa=(Wt, 1), sb=(a["embed"], 1))
        return t

    def text_backward(self, dt, ids_dev, eot_rows, N, L, tok_sort=None):
        a = self.arch
        Wt, E, M = a["text_width"], a["embed"], N * L
        lnf = self.buf["txt.lnf"]
        if self.requires_grad["text_projection"]:  # dproj[Wt,E] += lnf^T dt
            K.gemm_small(lnf, dt, self.P.g("text_projection"), M=Wt, N=E, K=N, sa=(1, Wt), sb=(E, 1), accumulate=True)
        dlnf = self._f("txt.dlnf", (N, Wt))
        K.gemm_small(dt, self.P.p("text_projection"), dlnf, M=N, N=Wt, K=E, sa=(E, 1), sb=(1, E))
        pruned = self.text_used_rows_only
        if pruned:  # the gradient of the last block's output exists at the EOT rows only: [N, Wt]
            dx, dxb = self._f("txt.dx_r", (N, Wt)), self._b("txt.dxb_r", (N, Wt))
            self._ln_bwd(dlnf, self.buf[f"txt{a['text_layers'] - 1}.xo_r"], "text_ln_final", "txt.lnf", dx, dx_bf16=dxb)
        else:
            dx = self._f("txt.dxA", (M, Wt), zero=True)
            dxb = self._b("txt.dxbA", (M, Wt), zero=True)
            self._ln_bwd(dlnf, self.buf[f"txt.x{a['text_layers']}"], "text_ln_final", "txt.lnf", dx, dx_bf16=dxb, rows=eot_rows)
        for l in reversed(range(a["text_layers"])):
            nx = "B" if (a["text_layers"] - l) % 2 == 1 else "A"
            dxi = self._f("txt.dx" + nx, (M, Wt))
            dxbi = self._b("txt.dxb" + nx, (M, Wt))
            if pruned and l == a["text_layers"] - 1:
                ht, hd = a["text_heads"], Wt // a["text_heads"]
                rows64, pos = self._text_eot  # the forward's tensors (the batch's when it came through prepare_batch)
                self._used_rows_bwd(f"text_model.resblocks.{l}.", _TEXT_NAMES, self.buf[f"txt.x{l}"], dx, dxb, dxi, dxbi, f"txt{l}",
                                    "txt.s", M, Wt, ht, eot_rows, a["act"],
                                    lambda qkv, datt, att, lse, delta, dqkv: K.attn_bwd_rowq(
                                        qkv, pos, datt, att, lse, delta, dqkv, B=N, heads=ht, S=L, head_dim=hd))
                dx, dxb = dxi, dxbi
                continue
            self._block_bwd(f"text_model.resblocks.{l}.", _TEXT_NAMES, self.buf[f"txt.x{l}"], dx, dxb, dxi, dxbi, f"txt{l}",
                            M, Wt, a["text_heads"], N, L, True, a["act"], "txt.s")
            dx, dxb = dxi, dxbi
        if self.requires_grad["text_token_embedding.weight"] or self.requires_grad["text_positional_embedding"]:
            K.text_embed_bwd(dx, ids_dev, self.P.g("text_token_embedding.weight"), self.P.g("text_positional_embedding"),
                             N=N, L=L, tok_sort=tok_sort)

    # ------------------------------------------------------------------ video tower
    def _attn_q8(self, M, W, name, T, n, persistent=False, consumer=None):
        """the divided-attention kernels write the per-tensor e4m3 copy of their result themselves (tensor mode, fused geometries;
        arch["fp8_attn_copies"] = False keeps the quantiser pass: same bytes, tests/test_model_gpu.py::test_fp8_wgrad_path)"""
        if not (self.fp8_wgrad and self._f8_tensor_mode and self.arch.get("fp8_attn_copies", True)) or n + 1 > 112 or T + 1 > 32:
            return {}, None
        q, sc, kw = self._q8(M, W, name, persistent=persistent)
        out = dict(q8out=q, q8_scale=sc, q8_amax=kw["amax"])
        # arch["fp8_q8_only"]: the e4m3 bytes are the kernels' only output for the patch rows -- every consumer of the attention
        # output (projection forward / weight gradient) and of the attention input gradient (qkv input / weight / bias gradient)
        # multiplies the e4m3 copy; the CLS rows, which the CLS-row delta, the hybrid stream's fix-up GEMM and the merge / finalize
        # kernels touch, keep both forms
        if self.fp8_q8_only and consumer is not None and self.requires_grad[consumer]:
            out["q8_only"] = True
        return out, (q, sc)

    def _st_attention_fwd(self, qkv, att, lse, mode, B, T, n, q8_for=None):
        h, S = self.arch["heads"], 1 + T * n
        ws = self._f("vit.clsws", (B * h * max(T, -(-n // 28)) * (self.dh + 2),))
        kw8 = {}
        if q8_for is not None and q8_for in self.P.w8:
            kw8, nxt = self._attn_q8(B * S, att.shape[1], "x." + q8_for, T, n, persistent=True, consumer=q8_for)
            if nxt is not None:
                self._x8_ready[q8_for] = nxt
        K.attn_fwd_divided(mode, qkv, att, lse, ws, B=B, heads=h, S=S, T=T, n=n, head_dim=self.dh, **kw8)

    def _st_attention_bwd(self, qkv, att, datt, lse, dqkv, mode, B, T, n, scr, q8_for=None):
        h, S = self.arch["heads"], 1 + T * n
        M = B * S
        kw8 = {}
        if q8_for is not None and q8_for in self.P.w8t:
            kw8, nxt = self._attn_q8(M, dqkv.shape[1], "dy." + q8_for, T, n, consumer=q8_for if q8_for in self._x8 else None)
            if nxt is not None:
                self._dy8_ready[q8_for] = nxt
        delta = self._f(scr + ".delta", (M, h))
        hd = self.dh
        # one partial of the CLS token's dK / dV / dQ per block of the fused kernels, added in order (no atomics)
        cls_acc = self._f(scr + ".clsacc", (B, h, max(T, -(-n // 28)), 3, hd))
        K.attn_bwd(mode, qkv, datt, att, lse, delta, dqkv, B=B, heads=h, S=S, T=T, n=n, cls_acc=cls_acc, head_dim=hd, **kw8)

    def video_forward(self, video, keep_dev, B, T, vid_rows=None):
        a = self.arch
        W, E, p = a["width"], a["embed"], a["patch"]
        n = keep_dev.shape[1]
        S = 1 + T * n
        M, Mp, Kp = B * S, B * T * n, self.P.conv_kpad
        cols = self._b("vit.im2col", (Mp, Kp))
        if video.dtype == torch.uint8:
            K.patch_gather_u8(video, keep_dev, cols, B=B, T=T, n=n, img=a["image"], patch=p, crop=self.ctx.get("crop"),
                              resize=self.ctx.get("resize"))
        else:
            K.patch_gather(video, keep_dev, cols, B=B, T=T, n=n, img=a["image"], patch=p)
        pe = self._f("vit.patch", (Mp, W))
        K.gemm_nt(cols, self.P.w_conv(), pe, M=Mp)
        tok = self._f("vit.tok", (M, W))
        K.vit_assemble(pe, self.P.p("video_model.class_embedding"), self.P.p("video_model.positional_embedding"),
                       self.P.p("video_model.temporal_embedding"), keep_dev, tok, B=B, T=T, n=n)
        lowres = self.bf16_residual
        xbuf = self._b if lowres else self._f   # the residual stream's buffers: bf16 or fp32
        x = xbuf("vit.x0", (M, W))
        self._ln(tok, "video_model.ln_pre", 1e-5, x, "vit.lnpre")
        cls = self.cls32
        if vid_rows is None:
            vid_rows = (torch.arange(B, device=self.dev) * S).to(torch.int32)
        xc = None
        if cls:  # the hybrid stream's CLS rows start as ln_pre's fp32 output on those rows
            xc = self._f("vit.xc0", (B, W))
            self._ln(tok, "video_model.ln_pre", 1e-5, xc, "vit.lnpre_c", rows=vid_rows)
        for l in range(a["layers"]):
            pre, tg = f"video_model.transformer.resblocks.{l}.", f"vit{l}"
            ln3 = self._b(tg + ".ln3", (M, W))
            a8 = self._ln(x, pre + "ln_3", 1e-5, ln3, tg + ".ln3", fp8_for=pre + "timeattn.qkv.weight", cls_x=xc)
            qkv_t = self._b(tg + ".qkv_t", (M, 3 * W))
            self._lin(ln3, pre + "timeattn.qkv.weight", pre + "timeattn.qkv.bias", qkv_t, M, a8=a8)
            att_t, lse_t = self._b(tg + ".att_t", (M, W)), self._f(tg + ".lse_t", (M, a["heads"]))
            self._st_attention_fwd(qkv_t, att_t, lse_t, "time", B, T, n, q8_for=pre + "timeattn.proj.weight")
            # the time residual only feeds ln_1 (the space branch restarts from x, video_encoder_ViT_B_16.py:121): bf16
            t_res = self._b(tg + ".t_res", (M, W))
            self._lin(att_t, pre + "timeattn.proj.weight", pre + "timeattn.proj.bias", t_res, M, residual=x)
            ln1 = self._b(tg + ".ln1", (M, W))
            a8 = self._ln(t_res, pre + "ln_1", 1e-5, ln1, tg + ".ln1", fp8_for=pre + "attn.qkv.weight")
            qkv_s = self._b(tg + ".qkv_s", (M, 3 * W))
            self._lin(ln1, pre + "attn.qkv.weight", pre + "attn.qkv.bias", qkv_s, M, a8=a8)
            att_s, lse_s = self._b(tg + ".att_s", (M, W)), self._f(tg + ".lse_s", (M, a["heads"]))
            self._st_attention_fwd(qkv_s, att_s, lse_s, "space", B, T, n, q8_for=pre + "attn.proj.weight")
            s_res = xbuf(tg + ".s_res", (M, W))  # residual from the block INPUT x (video_encoder_ViT_B_16.py:121)
            self._lin(att_s, pre + "attn.proj.weight", pre + "attn.proj.bias", s_res, M, residual=x)
            s_res_c = None
            if cls:
                s_res_c = self._f(tg + ".s_res_c", (B, W))
                self._cls_lin(att_s, S, pre + "attn.proj.weight", pre + "attn.proj.bias", xc, s_res_c)
            ln2 = self._b(tg + ".ln2", (M, W))
            a8 = self._ln(s_res, pre + "ln_2", 1e-5, ln2, tg + ".ln2", fp8_for=pre + "mlp.c_fc.weight", cls_x=s_res_c)
            h, act = self._b(tg + ".h", (M, 4 * W)), self._b(tg + ".a", (M, 4 * W))
            self._lin(ln2, pre + "mlp.c_fc.weight", pre + "mlp.c_fc.bias", act, M, act=a["act"], preact=h, a8=a8,
                      q8_for=pre + "mlp.c_proj.weight")
            xo = xbuf(f"vit.x{l + 1}", (M, W))
            self._lin(act, pre + "mlp.c_proj.weight", pre + "mlp.c_proj.bias", xo, M, residual=s_res)
            if cls:
                xcn = self._f(f"vit.xc{l + 1}", (B, W))
                self._cls_lin(act, S, pre + "mlp.c_proj.weight", pre + "mlp.c_proj.bias", s_res_c, xcn)
                xc = xcn
            x = xo
        out = self._f("vit.out", (M, E))
        if not self.pooled_tail:  # B models: ln_post on every token, all S projected rows feed the sort head
            lnp = self._b("vit.lnpost", (M, W))
            self._ln(x, "video_model.ln_post", 1e-5, lnp, "vit.lnpost", cls_x=xc)
            K.gemm_nt(lnp, self.P.wt("video_model.proj"), out, M=M)  # x @ proj, proj stored [W,E]
            return out, None
        # H/14 (video_encoder_ViT_H_14.py:472-484): pooled = ln_post(CLS) @ proj in fp32; the patch tokens are projected
        # WITHOUT ln_post (row 0 of each clip is computed too but never read: sort_assemble takes rows 1..S-1)
        if lowres:
            xb = self.buf["vit.xlast_b"] = x  # the stream is bf16 already
        else:
            xb = self._b("vit.xlast_b", (M, W))
            K.cast_f32_bf16(x, xb)
        K.gemm_nt(xb, self.P.wt("video_model.proj"), out, M=M)
        lnc = self._f("vit.lnpost_cls", (B, W))
        if cls:  # ln_post on the exact CLS rows of the hybrid stream
            self.buf["vit.xcls32"] = xc
            self._ln(xc, "video_model.ln_post", 1e-5, lnc, "vit.lnpost")
        elif lowres:  # ln_post on the B CLS rows in fp32: an fp32 copy of those rows (plumbing on [B, W])
            xc, xcb = self._f("vit.xcls32", (B, W)), self._b("vit.xcls16", (B, W))
            K.rows_move("gather", vid_rows, full_bf16=x, packed_f32=xc, packed_bf16=xcb)
            self._ln(xc, "video_model.ln_post", 1e-5, lnc, "vit.lnpost")
        else:
            self._ln(x, "video_model.ln_post", 1e-5, lnc, "vit.lnpost", rows=vid_rows)
        pooled = self._f("vit.pooled", (B, E))
        K.gemm_small(lnc, self.P.p("video_model.proj"), pooled, M=B, N=E, K=W, sa=(W, 1), sb=(E, 1))
        return out, pooled

    def video_backward(self, dout_b, keep_dev, B, T, d_pooled=None):
        """dout_b: bf16 [B*S, E] grad of the projected tokens (B models: CLS rows carry the embedding grad; H/14: None
        when there is no sorting loss); d_pooled: fp32 [B, E] grad of the pooled embedding (H/14 only)."""
        a, B_ = self.arch, self.buf
        W, E, p = a["width"], a["embed"], a["patch"]
        n = keep_dev.shape[1]
        S = 1 + T * n
        M, Mp = B * S, B * T * n
        rg = self.requires_grad
        dln = self._b("vit.s.dln", (M, W))
        # the residual-stream gradient in bf16 (self.bf16_grad_stream, see __init__): measured alone +1.4 % (the three LayerNorm
        # backwards of a block move 3.2 instead of 4.6 GB) at 2.4x the error of the embedding-side gradients, which sit behind
        # ln_pre's cancellation (5.8 % instead of 2.4 % rel-L2; profiles/r03_bf16_streams_ab.txt)
        lowp = self.bf16_grad_stream
        lowres = self.bf16_residual
        cls = self.cls32   # hybrid stream: the CLS rows of the gradient chain in fp32 side arrays dxc / dsrc [B, W]
        dxc = dsrc = None
        dxb = self._b("vit.dxbA", (M, W))
        dx = None if (lowp and not self.pooled_tail) else self._f("vit.dxA", (M, W))
        if not self.pooled_tail:
            if rg["video_model.proj"]:  # dproj[W,E] += lnpost^T dout
                K.gemm_tn(B_["vit.lnpost"], dout_b, self.P.g("video_model.proj"), M=M, accumulate=True)
            K.gemm_nt(dout_b, self.P.w("video_model.proj"), dln, M=M)
            if cls:
                dxc = self._f("vit.dxcA", (B, W))
            dxb8 = self._ln_bwd(dln, B_[f"vit.x{a['layers']}"], "video_model.ln_post", "vit.lnpost", dx, dx_bf16=dxb,
                                fp8_for=f"video_model.transformer.resblocks.{a['layers'] - 1}.mlp.c_proj.weight",
                                cls_x=B_[f"vit.xc{a['layers']}"] if cls else None, cls_dx=dxc)
        else:
            dxb8 = None
            # patch-token branch (no LN): dproj += x^T dout, dx = dout proj^T (CLS rows of dout are zero)
            if dout_b is not None:
                if rg["video_model.proj"]:
                    K.gemm_tn(B_["vit.xlast_b"], dout_b, self.P.g("video_model.proj"), M=M, accumulate=True)
                K.gemm_nt(dout_b, self.P.w("video_model.proj"), dx, M=M)
            else:
                K.zero_(dx)
            # pooled branch, fp32: dproj += lnc^T d_pooled, d_lnc = d_pooled proj^T, ln_post backward on the CLS rows only
            lnc = B_["vit.lnpost_cls"]
            if rg["video_model.proj"]:
                K.gemm_small(lnc, d_pooled, self.P.g("video_model.proj"), M=W, N=E, K=B, sa=(1, W), sb=(E, 1), accumulate=True)
            dlnc = self._f("vit.s.dlnc", (B, W))
            K.gemm_small(d_pooled, self.P.p("video_model.proj"), dlnc, M=B, N=W, K=E, sa=(E, 1), sb=(1, E))
            if lowres:  # ln_post backward on the fp32 copy of the CLS rows, added into those rows of dx
                # (hybrid stream: those rows of dx are zero -- no gradient enters the CLS rows through the patch-token branch -- so
                # the LayerNorm's result IS the fp32 CLS chain's first value)
                dxc = self._f("vit.dxcA" if cls else "vit.s.dxcls", (B, W))
                self._ln_bwd(dlnc, B_["vit.xcls32"], "video_model.ln_post", "vit.lnpost", dxc)
                K.rows_move("scatter_add", self.ctx["vid_rows"], full_f32=dx, packed_f32=dxc)
            else:
                self._ln_bwd(dlnc, B_[f"vit.x{a['layers']}"], "video_model.ln_post", "vit.lnpost", dx, res1=dx,
                             rows=self.ctx["vid_rows"])
            K.cast_f32_bf16(dx, dxb)
        datt = self._b("vit.s.datt", (M, W))
        dsr = None if lowp else self._f("vit.s.dsres", (M, W))
        # the weight gradients of the blocks on the side stream: the output gradients they read (dh, dsrb, dqkv of the space and of
        # the time branch, dtrb, and the block's own dxb) then live in buffers of the layer's parity, and the chain joins the side
        # stream's work of layer l + 1 before layer l overwrites the first of them (its ln_3 backward writes the dxb that layer
        # l + 1 read) -- the side stream may lag one layer behind, never two
        # (not with e4m3 weight gradients in tensor mode: those never take the side branch of _lin_bwd)
        side = self._wgrad_side(M) and not (self.fp8_wgrad and self._f8_tensor_mode)
        grouped = self._tn_grouped_on(M)
        side_done = {}
        cur = torch.cuda.current_stream(self.dev)
        if side:
            self._side()  # the stream exists before anything records on it or joins it (all weights frozen: no launch creates it)
        for l in reversed(range(a["layers"])):
            pre, tg = f"video_model.transformer.resblocks.{l}.", f"vit{l}"
            x_in = B_[f"vit.x{l}"]
            par = str(l % 2) if side else ""
            defer = [] if grouped else None
            dh = self._b("vit.s.dh" + par, (M, 4 * W))
            dsrb = self._b("vit.s.dsresb" + par, (M, W))
            dtrb = self._b("vit.s.dtresb" + par, (M, W))
            dqkv = self._b("vit.s.dqkv" + par, (M, 3 * W))
            dqkv_t = self._b("vit.s.dqkv_t" + par, (M, 3 * W)) if (side or grouped) else dqkv
            # (e4m3 input gradients: the LayerNorm backward that produces an output gradient also writes its e4m3 copy, d*8)
            self._lin_bwd(dxb, B_[tg + ".a"], pre + "mlp.c_proj.weight", pre + "mlp.c_proj.bias", dh, M, dy8=dxb8, side=side, defer=defer,
                          q8_for=pre + "mlp.c_fc.weight", gate_h=B_[tg + ".h"], gate_act=a["act"])
            self._lin_bwd(dh, B_[tg + ".ln2"], pre + "mlp.c_fc.weight", pre + "mlp.c_fc.bias", dln, M, side=side, defer=defer)
            if cls:
                dsrc = self._f("vit.s.dsrc", (B, W))
            dsrb8 = self._ln_bwd(dln, B_[tg + ".s_res"], pre + "ln_2", tg + ".ln2", dsr, dx_bf16=dsrb, res1=dxb if lowp else dx,
                                 fp8_for=pre + "attn.proj.weight", cls_x=B_[tg + ".s_res_c"] if cls else None,
                                 cls_res1=dxc if cls else None, cls_dx=dsrc)
            # spatial attention branch
            self._lin_bwd(dsrb, B_[tg + ".att_s"], pre + "attn.proj.weight", pre + "attn.proj.bias", datt, M, dy8=dsrb8, side=side,
                          defer=defer)
            self._st_attention_bwd(B_[tg + ".qkv_s"], B_[tg + ".att_s"], datt, B_[tg + ".lse_s"], dqkv, "space", B, T, n, "vit.s",
                                   q8_for=pre + "attn.qkv.weight")
            self._lin_bwd(dqkv, B_[tg + ".ln1"], pre + "attn.qkv.weight", pre + "attn.qkv.bias", dln, M, side=side, defer=defer)
            # the time-residual gradient is a side branch (t_res only feeds ln_1): it lives in bf16 only -- as the operand of
            # the timeattn.proj GEMMs and as the bf16 residual term of the ln_3 backward
            dtrb8 = self._ln_bwd(dln, B_[tg + ".t_res"], pre + "ln_1", tg + ".ln1", None, dx_bf16=dtrb,
                                 fp8_for=pre + "timeattn.proj.weight")
            # temporal attention branch
            self._lin_bwd(dtrb, B_[tg + ".att_t"], pre + "timeattn.proj.weight", pre + "timeattn.proj.bias", datt, M, dy8=dtrb8,
                          side=side, defer=defer)
            self._st_attention_bwd(B_[tg + ".qkv_t"], B_[tg + ".att_t"], datt, B_[tg + ".lse_t"], dqkv_t, "time", B, T, n, "vit.s",
                                   q8_for=pre + "timeattn.qkv.weight")
            self._lin_bwd(dqkv_t, B_[tg + ".ln3"], pre + "timeattn.qkv.weight", pre + "timeattn.qkv.bias", dln, M, side=side, defer=defer)
            if grouped:  # the block's six weight gradients (+ bias gradients) in one launch, their partials in one reduce launch:
                self._tn_group_run(l, defer)  # every output gradient they read is still intact (dqkv of the two branches apart)
            if side:
                side_done[l] = torch.cuda.Event()
                side_done[l].record(self._wg_stream)
                if l + 1 in side_done:  # layer l + 1's weight gradients are final: its dxb may be overwritten, its range may travel
                    cur.wait_event(side_done.pop(l + 1))
                    self._ready(f"video_model.transformer.resblocks.{l + 1}.")
            nx = "B" if (a["layers"] - l) % 2 == 1 else "A"
            dxbi = self._b("vit.dxb" + nx, (M, W))
            dxi = None if lowp else self._f("vit.dx" + nx, (M, W))
            # x feeds ln_3, the time residual and the space residual
            dxci = self._f("vit.dxc" + nx, (B, W)) if cls else None
            dxb8 = self._ln_bwd(dln, x_in, pre + "ln_3", tg + ".ln3", dxi, dx_bf16=dxbi, res1=dsrb if lowp else dsr, res2=dtrb,
                                fp8_for=f"video_model.transformer.resblocks.{l - 1}.mlp.c_proj.weight" if l > 0 else None,
                                cls_x=B_[f"vit.xc{l}"] if cls else None, cls_res1=dsrc, cls_dx=dxci)
            dx, dxb, dxc = dxi, dxbi, dxci
            if not side:
                self._ready(pre)
        if side:
            cur.wait_stream(self._wg_stream)
            self._ready("video_model.transformer.resblocks.0.")
        dtok = self._f("vit.dtok", (M, W))
        self._ln_bwd(dxb if lowp else dx, B_["vit.tok"], "video_model.ln_pre", "vit.lnpre", dtok)
        dpatch = self._b("vit.dpatch", (Mp, W))
        K.vit_assemble_bwd(dtok, keep_dev, dpatch, self.P.g("video_model.class_embedding"),
                           self.P.g("video_model.positional_embedding"), self.P.g("video_model.temporal_embedding"),
                           B=B, T=T, n=n)
        if rg["video_model.conv1.weight"]:
            if self.P.conv_pad is None:
                K.gemm_tn(dpatch, B_["vit.im2col"], self.P.g2d("video_model.conv1.weight"), M=Mp, accumulate=True)
            else:  # K padded to 64: wgrad into a [W, Kpad] scratch, the 588 real columns are folded into the gradient
                scr = self._f("vit.s.dconv", (W, self.P.conv_kpad))
                K.gemm_tn(dpatch, B_["vit.im2col"], scr, M=Mp, accumulate=False)
                K.add_rows_f32(self.P.g2d("video_model.conv1.weight"), scr)
        self._ready("video_model.class_embedding", "video_model.positional_embedding", "video_model.proj",
                    "video_model.temporal_embedding", "video_model.conv1.", "video_model.ln_pre.")
        self._ready("video_model.ln_post.")

    # ------------------------------------------------------------------ sort head
    def sort_forward(self, out, text_before, B, S, NT):
        a = self.arch
        E, hs = self.sort_width, a["sort_heads"]
        off = 1 if self.pooled_tail else 0  # H/14 hands the sort head the patch tokens without CLS
        Sv = S - off
        So = Sv + NT
        Mo = B * So
        xs = self._f("srt.x0", (Mo, E))
        K.sort_assemble(out, text_before, self.P.p("pred_model.type_embed").view(2, E), xs, B=B, S=S, off=off, Sv=Sv, NT=NT)
        x = xs
        rows = self.ctx["sort_rows"]
        nf = self._f("srt.nf", (B * NT, E))
        last = a["sort_depth"] - 1
        for l in range(a["sort_depth"]):
            if l == last and self.sort_used_rows_only and NT <= 16:
                # the head reads the last block's output at the NT transcript rows only (sort_transformer.py:131-141)
                xr = self._sort_last_fwd(f"pred_model.blocks.{l}.", x, f"srt{l}", Mo, E, hs, B, So, NT)
                self._ln(xr, "pred_model.norm", 1e-6, nf, "srt.norm")
                break
            xo = self._f(f"srt.x{l + 1}", (Mo, E))
            self._block_fwd(f"pred_model.blocks.{l}.", _SORT_NAMES, x, xo, f"srt{l}", Mo, E, hs, B, So, False, "gelu", 1e-6)
            x = xo
        else:
            self._ln(x, "pred_model.norm", 1e-6, nf, "srt.norm", rows=rows)
        pred = self._f("srt.pred", (B * NT, a["n_trans"]))
        C = a["n_trans"]
        K.gemm_small(nf, self.P.p("pred_model.head.weight"), pred, M=B * NT, N=C, K=E, sa=(E, 1), sb=(1, E),
                     bias=self.P.p("pred_model.head.bias"))
        return pred

    # ---- the LAST block of the sort head on the rows the model uses.  SortTransformer.forward_features normalises and classifies
    # x[:, x_len:] only (v2/model/sort_transformer.py:131-141): of the last block's [B, Sv + NT, E] output the NT transcript rows of
    # every sample are read, the Sv video rows never are, and no gradient enters them.  Keys and values of the block's attention
    # still come from every token, so LayerNorm 1 and the qkv projection run on all rows; the attention output, the output
    # projection, the residual, LayerNorm 2 and the MLP are computed for the R = B * NT used rows, and in the backward dQ exists for
    # those rows only while dK / dV (and through them the gradient of every input row) are dense.  Same loss, same gradient for
    # every parameter and every input row as the dense evaluation (tests/test_model_gpu.py::test_sort_head_used_rows_only);
    # arch["sort_used_rows_only"] = False evaluates the block densely like the reference does.
    def _sort_last_fwd(self, pre, x_in, tag, Mo, E, heads, B, So, NT):
        return self._used_rows_fwd(pre, _SORT_NAMES, x_in, tag, Mo, E, heads, self.ctx["sort_rows"], "gelu", 1e-6,
                                   lambda qkv, att, lse: K.attn_fwd_tail(qkv, att, lse, B=B, heads=heads, S=So, nq=NT, head_dim=E // heads))

    def _sort_last_bwd(self, pre, x_in, dxr, dxbr, dx_in, dxb_in, tag, Mo, E, heads, B, So, NT):
        def attn_bwd(qkv, datt, att, lse, delta, dqkv):
            K.zero_cols_bf16(dqkv, E)  # dQ of the rows that are no queries (dK / dV are written for every row)
            K.attn_bwd_tail(qkv, datt, att, lse, delta, dqkv, B=B, heads=heads, S=So, nq=NT, head_dim=E // heads)
        self._used_rows_bwd(pre, _SORT_NAMES, x_in, dxr, dxbr, dx_in, dxb_in, tag, "srt.s", Mo, E, heads, self.ctx["sort_rows"],
                            "gelu", attn_bwd)

    # A pre-LN block whose output the model reads at R rows only (rows64: their token rows), and into whose other output rows no
    # gradient enters.  LayerNorm 1 and the qkv projection stay dense (keys / values of every token); the attention output
    # (attn_fwd: a query-restricted kernel writing the used rows of `att`), the output projection, the residual, LayerNorm 2 and the
    # MLP are computed for the R used rows; backward: dQ for those rows, dK / dV -- and through them the gradient of every input
    # row -- dense.  Returns the block output at the used rows, [R, Wd] fp32.
    def _used_rows_fwd(self, pre, nm, x_in, tag, M, Wd, heads, rows32, act, eps, attn_fwd):
        R = rows32.numel()
        ln1 = self._b(tag + ".ln1", (M, Wd))
        self._ln(x_in, pre + nm["ln1"], eps, ln1, tag + ".ln1")
        qkv = self._b(tag + ".qkv", (M, 3 * Wd))
        self._lin(ln1, pre + nm["qkv_w"], pre + nm["qkv_b"], qkv, M)
        att, lse = self._b(tag + ".att", (M, Wd)), self._f(tag + ".lse", (M, heads))
        attn_fwd(qkv, att, lse)
        att_r, x_r = self._b(tag + ".att_r", (R, Wd)), self._f(tag + ".x_r", (R, Wd))
        K.rows_move("gather", rows32, full_bf16=att, packed_bf16=att_r)   # (row gathers of R x Wd elements)
        K.rows_move("gather", rows32, full_f32=x_in, packed_f32=x_r)
        mid = self._f(tag + ".mid", (R, Wd))
        self._lin(att_r, pre + nm["o_w"], pre + nm["o_b"], mid, R, residual=x_r)
        ln2 = self._b(tag + ".ln2", (R, Wd))
        self._ln(mid, pre + nm["ln2"], eps, ln2, tag + ".ln2")
        h, hact = self._b(tag + ".h", (R, 4 * Wd)), self._b(tag + ".a", (R, 4 * Wd))
        self._lin(ln2, pre + nm["fc_w"], pre + nm["fc_b"], hact, R, act=act, preact=h)
        xo = self._f(tag + ".xo_r", (R, Wd))
        self._lin(hact, pre + nm["pj_w"], pre + nm["pj_b"], xo, R, residual=mid)
        return xo

    def _used_rows_bwd(self, pre, nm, x_in, dxr, dxbr, dx_in, dxb_in, tag, scr, M, Wd, heads, rows32, act, attn_bwd):
        """dxr / dxbr: fp32 / bf16 gradient of the block output at the R used rows; writes the gradient of every input row."""
        R, B_ = rows32.numel(), self.buf
        dh, dln = self._b(scr + ".dh_r", (R, 4 * Wd)), self._b(scr + ".dln_r", (R, Wd))
        self._lin_bwd(dxbr, B_[tag + ".a"], pre + nm["pj_w"], pre + nm["pj_b"], dh, R, gate_h=B_[tag + ".h"], gate_act=act)
        self._lin_bwd(dh, B_[tag + ".ln2"], pre + nm["fc_w"], pre + nm["fc_b"], dln, R)
        dmid, dmidb = self._f(scr + ".dmid_r", (R, Wd)), self._b(scr + ".dmidb_r", (R, Wd))
        self._ln_bwd(dln, B_[tag + ".mid"], pre + nm["ln2"], tag + ".ln2", dmid, dx_bf16=dmidb, res1=dxr)
        datt_r = self._b(scr + ".datt_r", (R, Wd))
        self._lin_bwd(dmidb, B_[tag + ".att_r"], pre + nm["o_w"], pre + nm["o_b"], datt_r, R)
        datt = self._b(scr + ".datt", (M, Wd))            # token-row indexed like the attention output; only the R rows are read
        K.rows_move("scatter", rows32, full_bf16=datt, packed_bf16=datt_r)
        dqkv = self._b(scr + ".dqkv", (M, 3 * Wd))
        delta = self._f(scr + ".delta", (M, heads))
        attn_bwd(B_[tag + ".qkv"], datt, B_[tag + ".att"], B_[tag + ".lse"], delta, dqkv)
        dlnf = self._b(scr + ".dln", (M, Wd))
        self._lin_bwd(dqkv, B_[tag + ".ln1"], pre + nm["qkv_w"], pre + nm["qkv_b"], dlnf, M)
        self._ln_bwd(dlnf, x_in, pre + nm["ln1"], tag + ".ln1", dx_in, dx_bf16=dxb_in)
        # the residual path: + dmid at the used rows (fp32 sum, bf16 copy refreshed for those rows)
        K.rows_move("scatter_add", rows32, full_f32=dx_in, full_bf16=dxb_in, packed_f32=dmid)

    def sort_backward(self, dpred, B, S, NT):
        """-> fp32 grad of the sort-head input xs [B*So, E]."""
        a, B_ = self.arch, self.buf
        E, hs, C = self.sort_width, a["sort_heads"], a["n_trans"]
        So = S - (1 if self.pooled_tail else 0) + NT
        Mo, R = B * So, B * NT
        nf = B_["srt.nf"]
        K.gemm_small(dpred, nf, self.P.g("pred_model.head.weight"), M=C, N=E, K=R, sa=(1, C), sb=(E, 1), accumulate=True)
        ones = self._ones(R)
        K.gemm_small(ones, dpred, self.P.g("pred_model.head.bias").view(1, C), M=1, N=C, K=R, sa=(0, 1), sb=(C, 1),
                     accumulate=True)
        dnf = self._f("srt.dnf", (R, E))
        K.gemm_small(dpred, self.P.p("pred_model.head.weight"), dnf, M=R, N=E, K=C, sa=(C, 1), sb=(E, 1))
        pruned = self.sort_used_rows_only and NT <= 16
        if pruned:  # gradient of the last block's output exists at the transcript rows only: [R, E]
            dx, dxb = self._f("srt.dx_r", (R, E)), self._b("srt.dxb_r", (R, E))
            self._ln_bwd(dnf, B_[f"srt{a['sort_depth'] - 1}.xo_r"], "pred_model.norm", "srt.norm", dx, dx_bf16=dxb)
        else:
            dx = self._f("srt.dxA", (Mo, E), zero=True)
            dxb = self._b("srt.dxbA", (Mo, E), zero=True)
            self._ln_bwd(dnf, B_[f"srt.x{a['sort_depth']}"], "pred_model.norm", "srt.norm", dx, dx_bf16=dxb, rows=self.ctx["sort_rows"])
        for l in reversed(range(a["sort_depth"])):
            nx = "B" if (a["sort_depth"] - l) % 2 == 1 else "A"
            dxi, dxbi = self._f("srt.dx" + nx, (Mo, E)), self._b("srt.dxb" + nx, (Mo, E))
            if pruned and l == a["sort_depth"] - 1:
                self._sort_last_bwd(f"pred_model.blocks.{l}.", B_[f"srt.x{l}"], dx, dxb, dxi, dxbi, f"srt{l}", Mo, E, hs, B, So, NT)
                dx, dxb = dxi, dxbi
                continue
            self._block_bwd(f"pred_model.blocks.{l}.", _SORT_NAMES, B_[f"srt.x{l}"], dx, dxb, dxi, dxbi, f"srt{l}", Mo, E,
                            hs, B, So, False, "gelu", "srt.s")
            dx, dxb = dxi, dxbi
        return dx

    # ------------------------------------------------------------------ whole model
    def _clip_to_device(self, video: torch.Tensor) -> torch.Tensor:
        """The clip tensor (the one large input: 925 MB of fp32 at 192 pairs) goes host -> device on a COPY STREAM of the engine:
        on the compute stream the copy would queue behind the previous step's kernels and run in front of this step's -- 20 ms
        (pinned) to 46 ms (pageable) of a 140 ms step with nothing else running (tools/bench_fed.py).  On its own stream it travels
        while the previous step computes; the compute stream waits for it by event, the allocator is told who uses the block."""
        if video.is_cuda or not torch.cuda.is_available():
            return video.to(self.dev)
        cs = getattr(self, "_copy_stream", None)
        if cs is None:
            cs = self._copy_stream = torch.cuda.Stream(device=self.dev)
        cur = torch.cuda.current_stream(self.dev)
        if torch.cuda.is_current_stream_capturing():
            return video.to(self.dev)
        with torch.cuda.stream(cs):
            d = video.to(self.dev, non_blocking=True)
        cur.wait_stream(cs)
        d.record_stream(cur)
        return d

    def prepare_batch(self, data: dict):
        """Host-side (plumbing): dtype/device normalisation of the reference batch dict."""
        a = self.arch
        video = data["video"]
        crop = resize = crop_cpu = None
        if video.dtype == torch.uint8:
            # uint8 wire format: [B, T, H0, W0, 3] frames as decoded + resized; crop / 255 / normalise
            # happen inside the patch gather.  data["crop"]: [B, 2] (top, left) of a random crop, absent = centre crop.
            if video.dim() == 4:
                video = video.unsqueeze(1)
            assert video.dim() == 5 and video.shape[-1] == 3, "uint8 video must be [B, T, H, W, 3]"
            video = self._clip_to_device(video).contiguous()
            if data.get("crop") is not None:
                crop_cpu = data["crop"].detach().to("cpu", torch.int32).contiguous()
            if data.get("resize") is not None:  # the frames are the decoder's pictures: Resize(size) happens inside the gather
                from .data_loader.transforms import resize_tables
                resize = resize_tables(video.shape[2], video.shape[3], int(data["resize"]), self.dev)
        else:
            if video.dim() == 4:
                video = video.unsqueeze(1)
            video = self._clip_to_device(video).to(torch.float32).contiguous()
        B, T = video.shape[:2]
        # the reference fails on these with an indexing / broadcasting error (temporal_embedding[:T], nn.Embedding, pos[keep_ind]); the
        # kernels would read or write past the tables instead, so the batch is checked where it enters
        if T > a["num_frames"]:
            raise ValueError(f"clips of {T} frames, the temporal embedding has {a['num_frames']} rows")
        ids = data["text"]
        ids_cpu = ids.detach().to("cpu", torch.int64)
        if ids_cpu.numel() == 0 or ids_cpu.shape[0] % B:
            raise ValueError(f"text: {tuple(ids_cpu.shape)} token rows for {B} clips (expected n_trans * B rows, clip-major)")
        if int(ids_cpu.min()) < 0 or int(ids_cpu.max()) >= a["vocab"] or ids_cpu.shape[1] > a["context"]:
            raise IndexError(f"token ids must lie in [0, {a['vocab']}) and captions within the context of {a['context']}")
        eot = ids_cpu.argmax(dim=-1)
        L = int(eot.max()) + 1
        N = ids_cpu.shape[0]
        NT = N // B
        # every small index tensor of the batch is made on the HOST and travels in one asynchronous copy (_stage_small): no copy of
        # this function is ordered behind the compute stream's running step, so the host prepares batch t + 1 while step t computes
        small = {"eot_rows": (torch.arange(N) * L + eot).to(torch.int32), "eot_pos": eot.to(torch.int32),
                 "ids": ids_cpu[:, :L].to(torch.int32).contiguous()}
        # rows sorted by token id: the token-embedding gradient is then an ordered sum per id instead of a scatter of atomics
        small["tok_order"], small["tok_seg"] = K.token_sort(ids_cpu[:, :L])
        keep = None
        if "keep_ind" not in data:
            # device-drawn tube mask: data["mask_seed"] + the global number of the batch's first sample
            # reproduce the draw whatever the batch split; the reference draws it in the dataset worker
            # (YTTemporal_dataset.py:207-213) and ships it with the batch
            if "mask_seed" not in data:
                raise KeyError("batch dict needs 'keep_ind' or 'mask_seed' (+ optional 'sample_offset')")
            keep = K.tube_mask(int(data["mask_seed"]), int(data.get("sample_offset", 0)), B, patches_per_frame(a),
                               n_keep(a), device=self.dev)
        else:
            kc = data["keep_ind"]
            if kc.dim() != 2 or kc.shape[0] not in (1, B):
                raise ValueError(f"keep_ind {tuple(kc.shape)}: expected [B, n_keep] (or [1, n_keep] for the whole batch)")
            if kc.numel() and (int(kc.min()) < 0 or int(kc.max()) >= patches_per_frame(a)):
                raise IndexError(f"keep_ind must index the {patches_per_frame(a)} patches of a frame")
            if kc.shape[0] == 1 and B > 1:  # one tube mask for the whole batch (the downstream scripts pass arange(n)[None])
                kc = kc.expand(B, -1)
            if kc.is_cuda:
                keep = kc.to(torch.int32).contiguous()
            else:
                small["keep"] = kc.to(torch.int32).contiguous()
        n = (keep if keep is not None else small["keep"]).shape[1]
        S = 1 + T * n
        Sv = S - 1 if self.pooled_tail else S
        So = Sv + NT
        small["sort_rows"] = (torch.arange(B)[:, None] * So + Sv + torch.arange(NT)[None, :]).reshape(-1).to(torch.int32)
        small["vid_rows"] = (torch.arange(B) * S).to(torch.int32)
        if crop_cpu is not None:
            small["crop"] = crop_cpu
        if "label" in data and NT != 1:
            small["labels"] = data["label"].detach().reshape(-1).to("cpu", torch.int32)
        d = self._stage_small(small)
        if keep is None:
            keep = d["keep"]
        return dict(video=video, crop=d.get("crop", crop), resize=resize, ids=d["ids"], tok_sort=(d["tok_order"], d["tok_seg"]),
                    eot_rows=d["eot_rows"], eot_index=(None, d["eot_pos"]), keep=keep, B=B, T=T, N=N, NT=NT, L=L, n=n, S=S,
                    sort_rows=d["sort_rows"], vid_rows=d["vid_rows"], labels=d.get("labels"))

    STAGE_SLOTS = 4

    def _stage_small(self, small: dict) -> dict:
        """{name: small host int tensor} -> {name: device tensor}: packed into ONE page-locked staging buffer (a ring of STAGE_SLOTS, each
        guarded by the event of its last copy) and moved by ONE asynchronous copy on the engine's copy stream; the compute stream
        waits for it by event.  (One pageable .to(device) per tensor is a synchronous copy queued behind the running step: the host
        would wait for the previous step to finish before it could even start preparing the next one.)"""
        offs, total = {}, 0
        for k, t in small.items():
            offs[k] = total
            total += (t.numel() * t.element_size() + 15) // 16 * 16
        total = max(total, 16)
        if not torch.cuda.is_available() or self.dev.type != "cuda":
            return {k: t.to(self.dev) for k, t in small.items()}
        ring = getattr(self, "_stage_ring", None)
        if ring is None:
            ring = self._stage_ring = dict(i=0, slots=[None] * self.STAGE_SLOTS)
        i = ring["i"] = (ring["i"] + 1) % self.STAGE_SLOTS
        slot = ring["slots"][i]
        if slot is None or slot[0].numel() < total:
            slot = ring["slots"][i] = [torch.empty(max(total, 1 << 16), dtype=torch.uint8, pin_memory=True), None]
        if slot[1] is not None:
            slot[1].synchronize()  # the copy that last read this staging buffer (STAGE_SLOTS batches ago: long done)
        host = slot[0]
        for k, t in small.items():
            nb = t.numel() * t.element_size()
            if nb:
                host[offs[k]:offs[k] + nb].view(t.dtype).copy_(t.reshape(-1))
        cs = getattr(self, "_copy_stream", None)
        if cs is None:
            cs = self._copy_stream = torch.cuda.Stream(device=self.dev)
        cur = torch.cuda.current_stream(self.dev)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            devbuf = host[:total].to(self.dev)
        else:
            with torch.cuda.stream(cs):
                devbuf = torch.empty(total, dtype=torch.uint8, device=self.dev)
                devbuf.copy_(host[:total], non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record(cs)
            cur.wait_stream(cs)
            devbuf.record_stream(cur)
        out = {}
        for k, t in small.items():
            nb = t.numel() * t.element_size()
            out[k] = devbuf[offs[k]:offs[k] + nb].view(t.dtype).view(t.shape)
        return out

    def forward(self, pb: dict):
        """-> (text_emb [B,E], video_emb [B,E], pred [B*NT, n_trans] | None); all fp32 workspace tensors."""
        a = self.arch
        self.ctx = pb
        self._tick += 1  # (workspace: a buffer whose shape changes from here on belongs to a new step, see _b)
        B, T, N, NT, L, S, E = pb["B"], pb["T"], pb["N"], pb["NT"], pb["L"], pb["S"], a["embed"]
        text_emb = self._f("mdl.text_emb", (B, E))
        text_before = self._f("mdl.text_before", (B, NT, E))
        with self._text_lane():  # (the text tower beside the ViT when self.text_side, in line otherwise)
            t = self.text_forward(pb["ids"], pb["eot_rows"], N, L, eot_index=pb.get("eot_index"))
            K.text_mean(t, text_emb, text_before, NT=NT, B=B)
        out, pooled = self.video_forward(pb["video"], pb["keep"], B, T, vid_rows=pb["vid_rows"])
        if pooled is None:
            video_emb = self._f("mdl.video_emb", (B, E))
            K.rows_gather(out, pb["vid_rows"], video_emb)
        else:
            video_emb = pooled
        self._text_join()
        if self.embeds_ready is not None:  # the embedding all-gather starts here and travels under the sort head's forward
            self.embeds_ready(text_emb, video_emb)
        pred = self.sort_forward(out, text_before, B, S, NT) if (NT != 1 and self.has_sort_head) else None
        return text_emb, video_emb, pred

    class _TextLane:
        """`with` block whose launches go to the text tower's stream (forked from the current stream's position) and use the side
        scratch lane; a no-op when the text tower runs in line"""

        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            e = self.eng
            if not e.text_side:
                return
            if e._txt_stream is None:
                e._txt_stream = torch.cuda.Stream(device=e.dev)
            cur = torch.cuda.current_stream(e.dev)
            e._txt_stream.wait_stream(cur)
            self.sc, self.ln = torch.cuda.stream(e._txt_stream), K.lane(1)
            self.sc.__enter__()
            self.ln.__enter__()
            e._txt_open = True

        def __exit__(self, *exc):
            if self.eng.text_side:
                self.ln.__exit__(*exc)
                self.sc.__exit__(*exc)

    def _text_lane(self):
        return Engine._TextLane(self)

    def _text_join(self):
        """the current stream waits for what the text lane has queued"""
        if self.text_side and getattr(self, "_txt_open", False):
            torch.cuda.current_stream(self.dev).wait_stream(self._txt_stream)
            self._txt_open = False

    def backward(self, d_text, d_video, d_pred):
        """Accumulates parameter gradients into the flat grad buffer (+=).  d_* are fp32 GPU tensors (or None)."""
        a, pb = self.arch, self.ctx
        B, T, N, NT, L, S, E = pb["B"], pb["T"], pb["N"], pb["NT"], pb["L"], pb["S"], a["embed"]
        # the text tower first: it shares nothing with the video / sort-head backward (the sort head sees DETACHED caption
        # embeddings, model_dist..B_16.py:69), and its gradient range (token embedding included) is the largest single
        # all-reduce of the step -- issued here it travels under the whole video backward instead of after it
        # ... which holds while the text tower runs IN LINE.  On its own stream (text_side) its backward finishes some 150 kernels after
        # the fork, and an all-reduce issued from that lane would sit at the head of the single collective stream until then with every
        # ViT range queued behind it: with an exchange step attached (world > 1 / range-wise optimizer) the text range is handed over
        # at the join instead, from the main stream, behind the ViT ranges.
        text_ready_at_join = False
        if d_text is not None:
            dt = self._f("mdl.dt", (N, E))
            with self._text_lane():  # beside the sort head's and the ViT's backward when self.text_side; joined at the end
                K.text_mean_bwd(d_text, dt, NT=NT, B=B)
                self.text_backward(dt, pb["ids"], pb["eot_rows"], N, L, tok_sort=pb.get("tok_sort"))
                if self.text_side and (self.grad_ready is not None or self.param_ready is not None):
                    text_ready_at_join = True
                else:
                    self._ready("text_")
        dout = self._b("mdl.dout", (B * S, E))
        off = 1 if self.pooled_tail else 0
        dv_cls = None if self.pooled_tail else d_video  # B models: the embedding IS the CLS row of the projected tokens
        if d_pred is not None:
            dxs = self.sort_backward(d_pred, B, S, NT)
            K.sort_assemble_bwd(dxs, dv_cls, dout, self.P.g("pred_model.type_embed").view(2, E), B=B, S=S, off=off,
                                Sv=S - off, NT=NT)
            self._ready("pred_model.")
        elif not self.pooled_tail:
            K.sort_assemble_bwd(None, dv_cls, dout, None, B=B, S=S, off=0, Sv=S, NT=NT)
        else:
            dout = None
        self.video_backward(dout, pb["keep"], B, T, d_pooled=d_video if self.pooled_tail else None)
        self._text_join()
        if text_ready_at_join:
            self._ready("text_")


class LossHead:
    """sim_matrix + NormSoftmaxLoss (+ sorting CE) forward AND backward on the GPU, fp32.

    model/model_dist_TVTSv2_ViT_B_16.py:119-127, model/loss.py:13-25, trainer/trainer.py:484-496."""

    def __init__(self, device, temperature=0.05):
        self.dev, self.temp = device, temperature
        self.buf: Dict[str, torch.Tensor] = {}

    def _f(self, name, shape, zero=False):
        t = self.buf.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape:
            t = self.buf[name] = torch.zeros(shape, dtype=torch.float32, device=self.dev)
        elif zero:
            K.zero_(t)
        return t

    def sim(self, a, b, eps=1e-8):
        G, E = a.shape
        an, bn = self._f("an", (G, E)), self._f("bn", (b.shape[0], E))
        ai, bi = self._f("ai", (G,)), self._f("bi", (b.shape[0],))
        K.l2norm_rows(a, an, ai, eps)
        K.l2norm_rows(b, bn, bi, eps)
        return an, bn, ai, bi

    def contrastive(self, video_all, text_all, need_grad=True):
        """-> (loss1 scalar tensor, d_video_all, d_text_all).  Rows of the similarity are videos (trainer.py:484)."""
        G, E = video_all.shape
        vn, tn, vi, ti = self.sim(video_all, text_all)
        x = self._f("x", (G, G))
        K.gemm_small(vn, tn, x, M=G, N=G, K=E, sa=(E, 1), sb=(1, E), alpha=1.0 / self.temp)
        lse = self._f("lse", (2 * G,))
        loss = self._f("loss1", (1,), zero=True)
        dx = self._f("dx", (G, G)) if need_grad else None
        K.infonce(x, lse, dx, loss)
        if not need_grad:
            return loss, None, None
        dvn, dtn = self._f("dvn", (G, E)), self._f("dtn", (G, E))
        K.gemm_small(dx, tn, dvn, M=G, N=E, K=G, sa=(G, 1), sb=(E, 1), alpha=1.0 / self.temp)
        K.gemm_small(dx, vn, dtn, M=G, N=E, K=G, sa=(1, G), sb=(E, 1), alpha=1.0 / self.temp)
        dv, dt = self._f("dv", (G, E)), self._f("dt", (G, E))
        K.l2norm_rows_bwd(dvn, vn, vi, dv)
        K.l2norm_rows_bwd(dtn, tn, ti, dt)
        return loss, dv, dt

    def sorting(self, pred, labels_i32, need_grad=True):
        """2 * CrossEntropy (trainer.py:487-492) -> (loss2, dpred)."""
        loss = self._f("loss2", (1,), zero=True)
        dp = self._f("dpred", pred.shape) if need_grad else None
        K.cross_entropy(pred, labels_i32, 2.0, dp, loss)
        return loss, dp
