"""Thin tensor-level wrappers over the C ABI (include/tvts_hip.h).

PyTorch is used for device memory and the current HIP stream only; every function here launches a
hand-written gfx950 kernel.  Tensors must live on the GPU; a CPU tensor raises (no fallback).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

GEMM_PROFILE = None  # bench.py: list collecting (kind, flops, start_event, end_event) per MFMA GEMM launch
HBM_PROFILE = None   # bench.py: list collecting (family, algorithmic bytes, start_event, end_event) per HBM-bound launch


class _hbm:
    """Brackets one launch of an HBM-bound kernel family with HIP events on the launch stream while bench.py's instrumented step
    collects them (HBM_PROFILE is a list); free otherwise.  nbytes = the ALGORITHMIC bytes of the launch: every operand and result
    crossing HBM once."""

    def __init__(self, family, nbytes):
        self.family, self.nbytes = family, nbytes

    def __enter__(self):
        if HBM_PROFILE is not None:
            self.e0, self.e1 = Event(), Event()
            self.e0.record()

    def __exit__(self, *exc):
        if HBM_PROFILE is not None and exc[0] is None:
            self.e1.record()
            HBM_PROFILE.append((self.family, float(self.nbytes() if callable(self.nbytes) else self.nbytes), self.e0, self.e1))
        return False


def _nb(*tensors_rows):
    """bytes of (tensor, rows) pairs: rows x row width x element size (None entries are skipped)"""
    return sum(r * t.shape[-1] * t.element_size() for t, r in tensors_rows if t is not None)

ACT = {"none": 0, None: 0, "quick_gelu": 1, "gelu": 2, "add": 3}  # "add": gate slot only (bf16 residual added to the result)
MODE = {"full": 0, "space": 1, "time": 2, "cls": 3}


class HipError(RuntimeError):
    pass


def _chk(rc: int, name: str):
    if rc != 0:
        raise HipError(f"{name} failed with code {rc}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("tvts_amd kernels need GPU tensors (no CPU path)")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor"
    return t.stride(0)


# ---- per-call dispatch options.  The C library keeps no state: every alternative kernel path is selected by the `opts` word of
# the call (include/tvts_hip.h, TVTS_GEMM_* / TVTS_TN_* / TVTS_ATTN_*).  Tests and benches that want a whole engine step on an
# alternative path wrap it in `with hip.options(nt_tile=256): ...`; the defaults below are what every call uses otherwise.
# The one product use: `nt_cus` -- the CU reservation of the persistent GEMM grid when world > 1 (tvts_amd/dist.py).
_DEFAULTS = dict(nt_tile=0, nt_cus=0, nt_f32_patch=False, nt_streamk=None, tn_streamk=None, fp8_k32=False, tn_tile=0, tn_splits=0, tn_early_dma=None, tn_a_fast=None,
                 attn_tr=True, attn_shared=True, attn_fused=True, attn_ablate=0)
_OPTS = dict(_DEFAULTS)
if os.environ.get("TVTS_NT_F32_PATCH") == "1":  # A/B switch: the fp32 LDS patch of rounds 1 - 5 for the plain NT forms (same bits)
    _OPTS["nt_f32_patch"] = True


class options:
    """Context manager: dispatch options of every call made inside the block (restored on exit, also on an exception)."""

    def __init__(self, **kw):
        bad = set(kw) - set(_DEFAULTS)
        if bad:
            raise KeyError(f"unknown dispatch option(s) {sorted(bad)}")
        self.kw = kw

    def __enter__(self):
        self.saved = dict(_OPTS)
        _OPTS.update(self.kw)
        return self

    def __exit__(self, *exc):
        _OPTS.clear()
        _OPTS.update(self.saved)
        return False


def set_default(**kw):
    """Process-wide default of a dispatch option (Python side only; dist.py reserves CUs for the RCCL kernels with nt_cus)."""
    bad = set(kw) - set(_DEFAULTS)
    if bad:
        raise KeyError(f"unknown dispatch option(s) {sorted(bad)}")
    _OPTS.update(kw)


def _tile_bits(t):
    # "ring" (/ "ring2" "ring3" "ring4"): the one-block-per-CU ring form of the 128-column NT kernel (/ 128, 192, 256 tile rows), "noring": never take it
    return {0: 0, None: 0, 128: 1, 256: 2, "ring": 1 | 128, "ring2": 1 | 128 | (1 << 16), "ring3": 1 | 128 | (2 << 16), "ring4": 1 | 128 | (3 << 16), "noring": 16384,
            "128noring": 1 | 16384}[t]


def _sk_bits(sk):
    return 0 if sk is None else (32 if sk else 64)  # TVTS_GEMM_STREAMK / TVTS_GEMM_NO_STREAMK


def nt_opts(tile=None, cus=None, fp8_k32=None, streamk=None, side_deriv=False):
    tile = _OPTS["nt_tile"] if tile is None else tile
    cus = _OPTS["nt_cus"] if cus is None else cus
    k32 = _OPTS["fp8_k32"] if fp8_k32 is None else fp8_k32
    streamk = _OPTS["nt_streamk"] if streamk is None else streamk
    return (_tile_bits(tile) | (4 if k32 else 0) | (((int(cus) // 8) & 63) << 8) | _sk_bits(streamk) | ((1 << 20) if side_deriv else 0)
            | ((1 << 22) if _OPTS["nt_f32_patch"] else 0))   # TVTS_GEMM_F32_PATCH


def tn_opts(tile=None, splits=None, early_dma=None, a_fast=None, streamk=None):
    streamk = _OPTS["tn_streamk"] if streamk is None else streamk
    tile = _OPTS["tn_tile"] if tile is None else tile
    splits = _OPTS["tn_splits"] if splits is None else splits
    early_dma = _OPTS["tn_early_dma"] if early_dma is None else early_dma
    a_fast = _OPTS["tn_a_fast"] if a_fast is None else a_fast
    o = _tile_bits(tile) | (int(splits) << 8) | _sk_bits(streamk)
    if early_dma is not None and not early_dma:
        o |= 4
    if a_fast is not None:
        o |= 16 if a_fast else 8
    return o


def attn_opts(tr=None, shared=None, fused=None, ablate=None, q8_only=False):
    tr = _OPTS["attn_tr"] if tr is None else tr
    shared = _OPTS["attn_shared"] if shared is None else shared
    fused = _OPTS["attn_fused"] if fused is None else fused
    ablate = _OPTS["attn_ablate"] if ablate is None else ablate
    # bit 7: with a q8 copy, the fused divided kernels do not store the bf16 patch rows (the e4m3 bytes are their only output)
    return (0 if tr else 1) | (0 if shared else 2) | (0 if fused else 4) | ((int(ablate) & 7) << 4) | (128 if q8_only else 0)


def gemm_nt_select(M, N, tile=None):
    return _lib.load().tvts_gemm_nt_select(M, N, nt_opts(tile=tile))


def gemm_tn_select(M, Na, Nb, tile=None):
    return _lib.load().tvts_gemm_tn_select(M, Na, Nb, tn_opts(tile=tile))


STREAMK_TAKEN = [0]  # launches that took the stream-K walk / the fused reduce under a forcing option (tests assert the path ran)
NT_WORKSPACE = {}  # per (device, lane): scratch of the stream-K walk of gemm_nt (arrival counters + fp32 partial tiles)


def _nt_workspace(dev):
    """zeroed once: every launch leaves the arrival counters at zero again (include/tvts_hip.h)"""
    key = (dev, current_lane())
    ws = NT_WORKSPACE.get(key)
    if ws is None:
        ws = NT_WORKSPACE[key] = torch.zeros(_lib.load().tvts_gemm_nt_workspace_bytes(), dtype=torch.uint8, device=dev)
    return ws


def gemm_nt(a, b, out, *, M=None, bias=None, residual=None, act=None, preact=None, gate_h=None, gate_act=None, tile=None, cus=None,
            streamk=None, workspace=True, side_deriv=False):
    """out[M,N] = [act'(gate_h) *] act(a[M,K] @ b[N,K]^T + bias) [+ residual]; a, b bf16; out bf16 or fp32.
    streamk: None = the entry point's own choice, True / False force / forbid the stream-K walk (needs the workspace)."""
    lib = _lib.load()
    ws = _nt_workspace(a.device) if workspace else None
    M = a.shape[0] if M is None else M
    N, K = b.shape
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape[1] == K
    assert out.shape[1] == N and out.dtype in (torch.bfloat16, torch.float32)
    if residual is not None and residual.dtype == torch.bfloat16:  # the bf16 residual stream: the gate slot used additively
        assert gate_h is None and act is None
        gate_h, gate_act, residual = residual, "add", None
    if GEMM_PROFILE is not None:
        ev0, ev1 = Event(), Event()
        ev0.record()
    # the instrumented step of bench.py: plain bf16 launches of the 256 x 256 kernel sample the shader clock inside the kernel
    # (TVTS_GEMM_CLOCK_SAMPLE: 4 x u64 in the last 32 bytes of the workspace, copied to the probe buffer behind the launch)
    cp = CLOCK_PROBE
    sample = (cp is not None and ws is not None and cp["i"] < cp["buf"].shape[0] and out.dtype == torch.bfloat16
              and residual is None and act is None and gate_h is None and M >= 65536)
    if sample:
        zero_(ws[-32:])
    def launch(sk):
        return lib.tvts_gemm_nt_bf16(_p(a), _ld(a), _p(b), _ld(b), M, N, K, _p(bias), _p(residual),
                                     _ld(residual) if residual is not None else 0, ACT[act], _p(preact),
                                     _ld(preact) if preact is not None else 0, _p(gate_h),
                                     _ld(gate_h) if gate_h is not None else 0, ACT[gate_act], _p(out), _ld(out),
                                     1 if out.dtype == torch.float32 else 0, _p(ws), ws.numel() if ws is not None else 0,
                                     nt_opts(tile, cus, streamk=sk, side_deriv=side_deriv) | ((1 << 21) if sample else 0), _stream())
    want_sk = _OPTS["nt_streamk"] if streamk is None else streamk
    rc = launch(streamk)
    if rc == -22 and want_sk and streamk is None:  # the process-wide option means "wherever the shape can take it"
        rc = launch(False)
    elif rc == 0 and want_sk:
        STREAMK_TAKEN[0] += 1
    _chk(rc, "tvts_gemm_nt_bf16")
    if GEMM_PROFILE is not None:
        ev1.record()
        GEMM_PROFILE.append(("gemm_nt", 2.0 * M * N * K, ev0, ev1, (M, N, K, "f32" if out.dtype == torch.float32 else "bf16", "res" if residual is not None else "", str(act or ""), "gate" if gate_h is not None else "")))
    if sample:
        cp["buf"][cp["i"]].copy_(ws[-32:].view(torch.int64))
        cp["shape"].append((M, N, K))
        cp["i"] += 1


def quantize_fp8(x, q=None, scale=None, amax=None, amax_given=False):
    """per-tensor e4m3 quantisation of a bf16 / fp32 matrix -> (q uint8 [rows, cols], scale float32[1]); x ~ q * scale.
    q / scale / amax may be preallocated (the engine's workspace)."""
    lib = _lib.load()
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.bfloat16, torch.float32)
    assert not (amax_given and amax is None)
    amax = torch.empty(1, dtype=torch.float32, device=x.device) if amax is None else amax
    scale = torch.empty(1, dtype=torch.float32, device=x.device) if scale is None else scale
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if q is None else q
    f32 = 1 if x.dtype == torch.float32 else 0
    if not amax_given:  # amax_given: `amax` already holds max |x| (values beyond it saturate at +-448)
        _chk(lib.tvts_amax(_p(x), f32, x.stride(0), x.shape[0], x.shape[1], _p(amax), _stream()), "tvts_amax")
    _chk(lib.tvts_quant_fp8(_p(x), f32, x.stride(0), x.shape[0], x.shape[1], _p(amax), _p(q), q.stride(0), _p(scale), _stream()),
         "tvts_quant_fp8")
    return q, scale


def quantize_fp8_multi_table(entries, device):
    """device table for quantize_fp8_multi.  entries: (w fp32 [rows, cols] contiguous, q uint8, wt bf16 [cols, rows] or None,
    qt uint8 or None, amax float32[1], scale float32[1], scale_t float32[1] or None)"""
    import numpy as np
    rec = np.zeros(len(entries), dtype=[("w", "<u8"), ("q", "<u8"), ("wt", "<u8"), ("qt", "<u8"), ("amax", "<u8"), ("scale", "<u8"),
                                        ("scale_t", "<u8"), ("rows", "<i4"), ("cols", "<i4")])
    for i, (w, q, wt, qt, am, sc, sct) in enumerate(entries):
        assert w.dtype == torch.float32 and w.is_contiguous() and q.is_contiguous() and w.numel() % 4 == 0
        assert wt is None or (wt.dtype == torch.bfloat16 and wt.is_contiguous() and qt.is_contiguous() and wt.numel() == w.numel())
        rows, cols = w.shape[0], w.numel() // w.shape[0]
        rec[i] = (w.data_ptr(), q.data_ptr(), 0 if wt is None else wt.data_ptr(), 0 if qt is None else qt.data_ptr(), am.data_ptr(),
                  sc.data_ptr(), 0 if sct is None else sct.data_ptr(), rows, cols)
    return torch.from_numpy(rec.view(np.uint8).copy()).to(device), len(entries)


def quantize_fp8_multi(table, n):
    """per-tensor e4m3 copies of all the weights of the table (and of their transposed shadows) in three launches"""
    _chk(_lib.load().tvts_quant_fp8_multi(_p(table), n, _stream()), "tvts_quant_fp8_multi")


def quantize_fp8_rows(x, q=None, row_scale=None, tscale=None, amax=None):
    """per-row (per-token) e4m3 quantisation of a bf16 matrix in one pass -> (q uint8 [rows, cols], row_scale float32[rows]);
    x[r] ~ q[r] * row_scale[r].  tscale (float32[1], device): every row under THAT scale instead (per-tensor, delayed scaling;
    row_scale is then optional and returned as None when absent); amax (float32[1]): receives max(amax, max |x|)."""
    lib = _lib.load()
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.bfloat16
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if q is None else q
    if row_scale is None and tscale is None:
        row_scale = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    assert row_scale is None or (row_scale.numel() >= x.shape[0] and row_scale.dtype == torch.float32)
    with _hbm("quant_fp8_rows", _nb((x, x.shape[0]), (q, x.shape[0]))):
        rc = lib.tvts_quant_fp8_rows(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(q), q.stride(0), _p(row_scale), _p(tscale), _p(amax),
                                     _stream())
    _chk(rc, "tvts_quant_fp8_rows")
    return q, row_scale


def fp8_update_scales(amax, scale):
    """scale[i] = amax[i] / 448 where amax[i] > 0, then amax[i] = 0: a step's running maxima become the next step's per-tensor scales"""
    assert amax.dtype == scale.dtype == torch.float32 and amax.numel() == scale.numel()
    _chk(_lib.load().tvts_fp8_update_scales(_p(amax), _p(scale), amax.numel(), _stream()), "tvts_fp8_update_scales")


def gemm_nt_fp8(a8, sa, b8, sb, out, *, bias=None, residual=None, act=None, preact=None, gate_h=None, gate_act=None, k32=None,
                cus=None, q8out=None, q8_scale=None, q8_amax=None, store_out=True, side_deriv=False):
    """out[M,N] = act(sa*sb * (a8[M,K] @ b8[N,K]^T) + bias) [+ residual]; a8 / b8 uint8 e4m3 bit patterns; sb float32[1]; sa
    float32[1] (one scale for the tensor) or float32[>= M] (one per row, quantize_fp8_rows); preact receives the bf16
    pre-activation like gemm_nt.  gate_h / gate_act: the input-gradient form, out = gate_act'(gate_h) * (sa*sb * (a8 @ b8^T)).
    k32: the 16x16x32 fp8 main loop instead of the K = 128 scaled MFMA (same products, same fp32 accumulation).
    store_out=False (with q8out): the bf16 result is not written, the e4m3 copy is the only output."""
    lib = _lib.load()
    assert store_out or q8out is not None
    po = _p(out) if store_out else None
    M, Kd = a8.shape
    N = b8.shape[0]
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and b8.shape[1] == Kd
    sa_rows = 0 if sa.numel() == 1 else 1
    assert sa_rows == 0 or sa.numel() >= M
    if GEMM_PROFILE is not None:
        ev0, ev1 = Event(), Event()
        ev0.record()
    if residual is not None and residual.dtype == torch.bfloat16:
        assert gate_h is None and act is None and preact is None
        gate_h, gate_act, residual = residual, "add", None
    if gate_h is not None:
        assert residual is None and act is None and preact is None and out.dtype == torch.bfloat16
        assert bias is None or gate_act == "add"
        rc = lib.tvts_gemm_nt_fp8_gate(_p(a8), a8.stride(0), _p(b8), b8.stride(0), M, N, Kd, _p(sa), sa_rows, _p(sb), _p(bias), _p(gate_h),
                                       _ld(gate_h), ACT[gate_act], po, _ld(out), _p(q8out), q8out.stride(0) if q8out is not None else 0, _p(q8_scale),
                                       _p(q8_amax), nt_opts(None, cus, k32, side_deriv=side_deriv), _stream())
        _chk(rc, "tvts_gemm_nt_fp8_gate")
        if GEMM_PROFILE is not None:
            ev1.record()
            GEMM_PROFILE.append(("gemm_nt_fp8", 2.0 * M * N * Kd, ev0, ev1, (M, N, Kd, "fp8", "", "", "gate")))
        return
    rc = lib.tvts_gemm_nt_fp8(_p(a8), a8.stride(0), _p(b8), b8.stride(0), M, N, Kd, _p(sa), sa_rows, _p(sb), _p(bias), _p(residual),
                              _ld(residual) if residual is not None else 0, ACT[act], _p(preact),
                              _ld(preact) if preact is not None else 0, po, _ld(out),
                              1 if out.dtype == torch.float32 else 0, _p(q8out), q8out.stride(0) if q8out is not None else 0,
                              _p(q8_scale), _p(q8_amax), nt_opts(None, cus, k32, side_deriv=side_deriv), _stream())
    _chk(rc, "tvts_gemm_nt_fp8")
    if GEMM_PROFILE is not None:
        ev1.record()
        GEMM_PROFILE.append(("gemm_nt_fp8", 2.0 * M * N * Kd, ev0, ev1, (M, N, Kd, "fp8", "res" if residual is not None else "", str(act or ""), "")))


# Scratch LANES: the shared scratch buffers of this module (split partials of the weight gradients, LayerNorm dgamma / dbeta partials,
# stream-K workspace) belong to ONE stream's launch order.  A second stream that runs kernels beside it (the text tower next to the
# ViT, Engine.text_side) selects its own set with `with K.lane(1):` -- per host thread, like torch's current stream.
import threading

_LANE = threading.local()


def current_lane() -> int:
    return getattr(_LANE, "i", 0)


class lane:
    def __init__(self, i: int):
        self.i = int(i)

    def __enter__(self):
        self.prev = current_lane()
        _LANE.i = self.i

    def __exit__(self, *exc):
        _LANE.i = self.prev


TN_WORKSPACE = {}  # per (device, lane): fp32 scratch tensor for the split partials of gemm_tn (allocated lazily, 256 MiB: 8 partials of the largest weight, H/14 mlp 1280 x 5120)


def _tn_workspace(dev):
    key = (dev, current_lane())
    ws = TN_WORKSPACE.get(key)
    if ws is None:
        ws = TN_WORKSPACE[key] = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)  # (every lane the same size: the split plan depends on it)
    return ws


def warm_scratch(dev, lanes=(0, 1)):
    """allocate every lane's shared scratch NOW (Engine.__init__): a buffer that is first touched inside a hipGraph capture would live
    in that graph's private pool and die with it, while this module keeps handing it out"""
    for i in lanes:
        with lane(i):
            _tn_counters(_tn_workspace(dev))
            _ln_workspace(dev)
            _nt_workspace(dev)


TN_COUNTERS = {}  # per workspace: zeroed arrival counters of the fused reduce (every launch leaves them at zero again)


def _tn_counters(ws):
    c = TN_COUNTERS.get(ws.data_ptr())
    if c is None:
        c = TN_COUNTERS[ws.data_ptr()] = torch.zeros(4096, dtype=torch.int32, device=ws.device)
    return c


def gemm_tn(p, q, out, *, M=None, accumulate=True, colsum=None, workspace=True, tile=None, splits=None, early_dma=None, a_fast=None,
            fused=None):
    """out[Na,Nb] (+)= p[M,Na]^T @ q[M,Nb]; p, q bf16; out fp32.  fused: None = the entry point's choice, True / False force /
    forbid the in-kernel reduce of the split partials (the last block of a tile adds them) against the separate reduce pass."""
    lib = _lib.load()
    M = p.shape[0] if M is None else M
    assert p.dtype == torch.bfloat16 and q.dtype == torch.bfloat16 and out.dtype == torch.float32
    # workspace: True = this process's shared scratch (one stream), a float32 tensor = the caller's (a second stream's own), False = none
    ws = workspace if isinstance(workspace, torch.Tensor) else (_tn_workspace(p.device) if workspace else None)
    cnt = _tn_counters(ws) if ws is not None else None
    if GEMM_PROFILE is not None:
        ev0, ev1 = Event(), Event()
        ev0.record()
    def launch(fu):
        return lib.tvts_gemm_tn_bf16(_p(p), _ld(p), _p(q), _ld(q), M, p.shape[1], q.shape[1], _p(out), _ld(out),
                                     1 if accumulate else 0, _p(colsum), _p(ws), ws.numel() if ws is not None else 0,
                                     _p(cnt), cnt.numel() if cnt is not None else 0,
                                     tn_opts(tile, splits, early_dma, a_fast, fu), _stream())
    want_fu = _OPTS["tn_streamk"] if fused is None else fused
    rc = launch(fused)
    if rc == -22 and want_fu and fused is None:
        rc = launch(False)
    elif rc == 0 and want_fu:
        STREAMK_TAKEN[0] += 1
    _chk(rc, "tvts_gemm_tn_bf16")
    if GEMM_PROFILE is not None:
        ev1.record()
        GEMM_PROFILE.append(("gemm_tn", 2.0 * M * p.shape[1] * q.shape[1], ev0, ev1, (M, p.shape[1], q.shape[1], "cs" if colsum is not None else "")))


class TnGroup:
    """The weight gradients of one group (a ViT block's six) launched together: tvts_gemm_tn_bf16_grouped.  Build it once with the
    problems -- dicts of the gemm_tn arguments p, q, out, M, accumulate, colsum -- and call run(); the device plan is uploaded by the
    first run -- one asynchronous copy from page-locked staging memory ON THE LAUNCH STREAM, in front of the kernels that read it --
    and re-used while the problems stay the same tensors (the engine's buffers are persistent)."""

    class _Rec(ctypes.Structure):
        _fields_ = [("P", ctypes.c_void_p), ("ldp", ctypes.c_int), ("Q", ctypes.c_void_p), ("ldq", ctypes.c_int), ("M", ctypes.c_int),
                    ("Na", ctypes.c_int), ("Nb", ctypes.c_int), ("out", ctypes.c_void_p), ("ldo", ctypes.c_int),
                    ("accumulate", ctypes.c_int), ("colsum", ctypes.c_void_p)]

    def __init__(self, problems, workspace, splits=0):
        lib = _lib.load()
        self.n = len(problems)
        self.keep = problems  # the tensors stay alive with the plan
        self.recs = (self._Rec * self.n)()
        for r, pr in zip(self.recs, problems):
            p, q, out = pr["p"], pr["q"], pr["out"]
            assert p.dtype == torch.bfloat16 and q.dtype == torch.bfloat16 and out.dtype == torch.float32
            r.P, r.ldp, r.Q, r.ldq = p.data_ptr(), _ld(p), q.data_ptr(), _ld(q)
            r.M, r.Na, r.Nb = int(pr.get("M") or p.shape[0]), p.shape[1], q.shape[1]
            r.out, r.ldo, r.accumulate = out.data_ptr(), _ld(out), 1 if pr.get("accumulate", True) else 0
            r.colsum = pr["colsum"].data_ptr() if pr.get("colsum") is not None else None
        self.key = tuple((r.P, r.Q, r.out, r.colsum, r.M, r.Na, r.Nb, r.ldp, r.ldq, r.ldo, r.accumulate) for r in self.recs)
        self.ws = workspace
        nbytes = lib.tvts_gemm_tn_grouped_table_bytes(self.n)
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=workspace.device)  # (no fill: a fill on another stream could land after the upload)
        self.table_host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)    # stays alive and untouched with the plan
        self.uploaded = False
        self.captured = False  # a hipGraph holds the addresses of table / table_host: the plan must outlive it (Engine never evicts it)
        self.opts = int(splits) << 8
        self.flops = sum(2.0 * r.M * r.Na * r.Nb for r in self.recs)

    def run(self):
        lib = _lib.load()
        if GEMM_PROFILE is not None:
            ev0, ev1 = Event(), Event()
            ev0.record()
        # An upload issued while the stream is being captured becomes a memcpy NODE of the graph: it runs at every replay, not now.  The
        # table then counts as uploaded for the replays only -- an eager run of the same plan (before or between replays) uploads again.
        capturing = torch.cuda.is_current_stream_capturing()
        rc = lib.tvts_gemm_tn_bf16_grouped(ctypes.cast(self.recs, ctypes.c_void_p), self.n, _p(self.table), ctypes.c_void_p(self.table_host.data_ptr()),
                                           self.table.numel(), 0 if (self.uploaded and not capturing) else 1, _p(self.ws), self.ws.numel(), self.opts, _stream())
        _chk(rc, "tvts_gemm_tn_bf16_grouped")
        if capturing:
            self.captured = True
        else:
            self.uploaded = True
        if GEMM_PROFILE is not None:
            ev1.record()
            r0 = self.recs[0]
            GEMM_PROFILE.append(("gemm_tn", self.flops, ev0, ev1, (r0.M, sum(r.Na * r.Nb for r in self.recs) // max(r0.Nb, 1), r0.Nb, "grouped")))


def gemm_tn_fp8(p8, sp, q8, sq, out, *, M=None, accumulate=True, colsum=None, workspace=True, splits=None):
    """out[Na,Nb] (+)= sp * sq * p8[M,Na]^T @ q8[M,Nb]; p8 / q8 uint8 e4m3 bytes under one scale per tensor (float32[1] each);
    colsum[a] += sp * sum_m p8[m,a] (the bias gradient from the same bytes)."""
    lib = _lib.load()
    M = p8.shape[0] if M is None else M
    assert p8.dtype == torch.uint8 and q8.dtype == torch.uint8 and out.dtype == torch.float32 and sp.numel() == 1 and sq.numel() == 1
    ws = workspace if isinstance(workspace, torch.Tensor) else (_tn_workspace(p8.device) if workspace else None)
    if GEMM_PROFILE is not None:
        ev0, ev1 = Event(), Event()
        ev0.record()
    rc = lib.tvts_gemm_tn_fp8(_p(p8), p8.stride(0), _p(q8), q8.stride(0), M, p8.shape[1], q8.shape[1], _p(sp), _p(sq), _p(out), _ld(out),
                              1 if accumulate else 0, _p(colsum), _p(ws), ws.numel() if ws is not None else 0, int(splits or 0) << 8, _stream())
    _chk(rc, "tvts_gemm_tn_fp8")
    if GEMM_PROFILE is not None:
        ev1.record()
        GEMM_PROFILE.append(("gemm_tn_fp8", 2.0 * M * p8.shape[1] * q8.shape[1], ev0, ev1, (M, p8.shape[1], q8.shape[1], "fp8")))


def gemm_small(a, b, out, *, M, N, K, sa, sb, alpha=1.0, bias=None, accumulate=False):
    """out[i,j] (+)= alpha * sum_k a[i*sa[0]+k*sa[1]] * b[k*sb[0]+j*sb[1]] + bias[j] (fp32)."""
    lib = _lib.load()
    assert a.dtype == b.dtype == out.dtype == torch.float32
    rc = lib.tvts_gemm_small_f32(_p(a), sa[0], sa[1], _p(b), sb[0], sb[1], M, N, K, alpha, _p(bias), _p(out),
                                 out.stride(0), 1 if accumulate else 0, _stream())
    _chk(rc, "tvts_gemm_small_f32")


def rows_linear(a, w, out, *, bias=None, residual=None):
    """out[R, N] (fp32) = residual + bias + a[R, K] @ w[N, K]^T for a FEW rows (a may be a strided row view: one row per clip)"""
    lib = _lib.load()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.float32 and a.stride(1) == 1
    R, Kd = a.shape
    N = w.shape[0]
    _chk(lib.tvts_rows_linear_bf16(_p(a), a.stride(0), _p(w), _ld(w), R, N, Kd, _p(bias), _p(residual),
                                   _ld(residual) if residual is not None else 0, _p(out), _ld(out), _stream()), "tvts_rows_linear_bf16")


def colsum(x, out, *, M=None):
    lib = _lib.load()
    M = x.shape[0] if M is None else M
    ws = _tn_workspace(x.device)
    _chk(lib.tvts_colsum_bf16(_p(x), _ld(x), M, x.shape[1], _p(out), _p(ws), ws.numel(), _stream()), "tvts_colsum_bf16")


def layernorm_fwd(x, gamma, beta, eps, y, mean=None, rstd=None, rows=None, M=None, q8=None, row_scale=None, tscale=None, amax=None,
                  cls_x=None, cls_period=0, refresh=True):
    """q8 (uint8 [M, W]) + row_scale (float32 [>= M]): also write the bf16 output as e4m3 bytes with one scale per row.
    y = None (with q8, bf16 x, W % 8 == 0, W <= 1536): the e4m3 bytes are the only output.
    cls_x (float32 [M / cls_period, W]) + cls_period: the hybrid residual stream -- x bf16, the rows r % cls_period == 0 are read
    from cls_x and (refresh) their bf16 rounding is written back into x."""
    lib = _lib.load()
    M = (rows.numel() if rows is not None else x.shape[0]) if M is None else M
    fam = "ln_fwd" if M >= 4096 else "ln_fwd_small"
    assert y is not None or q8 is not None
    ldy = _ld(y) if y is not None else 0
    if cls_x is not None:
        assert rows is None and x.dtype == torch.bfloat16 and (y is None or y.dtype == torch.bfloat16) and cls_x.dtype == torch.float32
        assert cls_x.is_contiguous() and cls_x.shape[1] == x.shape[1] and cls_x.shape[0] * cls_period == M
        with _hbm(fam, _nb((x, M), (y, M), (q8, M)) + 8 * M):
            rc = lib.tvts_layernorm_fwd_cls(_p(x), _ld(x), _p(cls_x), cls_period, _p(x) if refresh else None, _p(gamma), _p(beta), eps, M,
                                            x.shape[1], _p(y), ldy, _p(q8), q8.stride(0) if q8 is not None else 0, _p(row_scale), _p(tscale),
                                            _p(amax), _p(mean), _p(rstd), _stream())
        _chk(rc, "tvts_layernorm_fwd_cls")
        return
    if q8 is not None:
        assert (y is None or y.dtype == torch.bfloat16) and q8.dtype == torch.uint8 and (tscale is not None or (row_scale.dtype == torch.float32 and row_scale.numel() >= M))
        with _hbm(fam, _nb((x, M), (y, M), (q8, M)) + 12 * M):
            rc = lib.tvts_layernorm_fwd_fp8(_p(x), _ld(x), 1 if x.dtype == torch.bfloat16 else 0, _p(rows), _p(gamma), _p(beta), eps, M,
                                            x.shape[1], _p(y), ldy, _p(q8), q8.stride(0), _p(row_scale), _p(tscale), _p(amax), _p(mean), _p(rstd), _stream())
        _chk(rc, "tvts_layernorm_fwd_fp8")
        return
    with _hbm(fam, _nb((x, M), (y, M)) + 8 * M):
        rc = lib.tvts_layernorm_fwd(_p(x), _ld(x), 1 if x.dtype == torch.bfloat16 else 0, _p(rows), _p(gamma), _p(beta), eps, M, x.shape[1], _p(y), _ld(y),
                                    1 if y.dtype == torch.float32 else 0, _p(mean), _p(rstd), _stream())
    _chk(rc, "tvts_layernorm_fwd")


LN_WORKSPACE = {}  # per (device, lane): fp32 scratch for the per-block dgamma/dbeta partials of layernorm_bwd (1024 blocks x 2 x 1280)


def _ln_workspace(dev):
    key = (dev, current_lane())
    ws = LN_WORKSPACE.get(key)
    if ws is None:
        ws = LN_WORKSPACE[key] = torch.empty(1024 * 2 * 1280, dtype=torch.float32, device=dev)
    return ws


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, *, dx_bf16=None, res1=None, res2=None, dgamma=None, dbeta=None,
                  rows=None, M=None, workspace=True, q8=None, row_scale=None, tscale=None, amax=None,
                  cls_period=0, cls_x=None, cls_res1=None, cls_dx=None):
    """dx (fp32, may be None when only the bf16 copy is wanted) = LN backward [+ res1 (fp32 or bf16) + res2 (bf16)];
    dgamma / dbeta are ACCUMULATED (+=) from per-block partials in a shared scratch buffer.  q8 / row_scale: also the e4m3
    copy of dx_bf16 with one scale per row (bf16 dy, every row).
    cls_period > 0: the hybrid residual stream (bf16 dy / x / res1, bf16 output only) -- for the rows r % cls_period == 0 the input
    comes from cls_x, the stream gradient from cls_res1 (both float32 [M / cls_period, W], optional) and the result also goes to
    cls_dx in fp32 (optional)."""
    lib = _lib.load()
    ws = _ln_workspace(x.device) if (workspace and dgamma is not None) else None
    M = (rows.numel() if rows is not None else x.shape[0]) if M is None else M
    assert res2 is None or res2.dtype == torch.bfloat16
    fam = "ln_bwd" if M >= 4096 else "ln_bwd_small"
    nbytes = _nb((dy, M), (x, M), (res1, M), (res2, M), (dx, M), (dx_bf16, M), (q8, M)) + 8 * M
    if cls_period:
        assert rows is None and dx is None and dx_bf16 is not None and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
        assert res1 is None or res1.dtype == torch.bfloat16
        for t in (cls_x, cls_res1, cls_dx):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] * cls_period == M and t.shape[1] == x.shape[1])
        with _hbm(fam, nbytes):
            rc = lib.tvts_layernorm_bwd_cls(_p(dy), _ld(dy), _p(x), _ld(x), _p(cls_x), _p(cls_res1), _p(cls_dx), cls_period, _p(mean), _p(rstd),
                                            _p(gamma), _p(res1), _ld(res1) if res1 is not None else 0, _p(res2), _ld(res2) if res2 is not None else 0,
                                            M, x.shape[1], _p(dx_bf16), _ld(dx_bf16), _p(q8), q8.stride(0) if q8 is not None else 0, _p(row_scale),
                                            _p(tscale), _p(amax), _p(dgamma), _p(dbeta), _p(ws), ws.numel() if ws is not None else 0, _stream())
        _chk(rc, "tvts_layernorm_bwd_cls")
        return
    if q8 is not None:
        assert rows is None and dx_bf16 is not None and dy.dtype == torch.bfloat16 and (tscale is not None or (row_scale is not None and row_scale.numel() >= M))
        with _hbm(fam, nbytes):
            rc = lib.tvts_layernorm_bwd_fp8(_p(dy), _ld(dy), _p(x), _ld(x), 1 if x.dtype == torch.bfloat16 else 0, _p(mean), _p(rstd),
                                        _p(gamma), _p(res1), 1 if (res1 is not None and res1.dtype == torch.bfloat16) else 0,
                                        _ld(res1) if res1 is not None else 0, _p(res2),
                                        _ld(res2) if res2 is not None else 0, M, x.shape[1], _p(dx), _ld(dx) if dx is not None else 0,
                                            _p(dx_bf16), _ld(dx_bf16), _p(q8), q8.stride(0), _p(row_scale), _p(tscale), _p(amax), _p(dgamma), _p(dbeta), _p(ws),
                                            ws.numel() if ws is not None else 0, _stream())
        _chk(rc, "tvts_layernorm_bwd_fp8")
        return
    with _hbm(fam, nbytes):
        rc = lib.tvts_layernorm_bwd(_p(dy), _ld(dy), 1 if dy.dtype == torch.float32 else 0, _p(x), _ld(x),
                                1 if x.dtype == torch.bfloat16 else 0, _p(rows),
                                _p(mean), _p(rstd), _p(gamma), _p(res1), 1 if (res1 is not None and res1.dtype == torch.bfloat16) else 0,
                                _ld(res1) if res1 is not None else 0, _p(res2),
                                _ld(res2) if res2 is not None else 0, M, x.shape[1], _p(dx),
                                    _ld(dx) if dx is not None else 0, _p(dx_bf16), _ld(dx_bf16) if dx_bf16 is not None else 0,
                                    _p(dgamma), _p(dbeta), _p(ws), ws.numel() if ws is not None else 0, _stream())
    _chk(rc, "tvts_layernorm_bwd")


def _attn_fn(lib, name, head_dim):
    if head_dim == 64:
        return getattr(lib, "tvts_attn_" + name)
    if head_dim == 80:
        return getattr(lib, "tvts_attn80_" + name)
    raise HipError(f"attention kernels are built for head dim 64 and 80, not {head_dim}")


def attn_fwd(mode, qkv, out, lse2, *, B, heads, S, T=0, n=0, causal=False, head_dim=64, **opt):
    lib = _lib.load()
    M = B * S
    with _hbm("attn_fwd_" + mode, _nb((qkv, M), (out, M)) + 8 * M * heads):
        rc = _attn_fn(lib, "fwd", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, int(causal), _p(out), _ld(out), _p(lse2),
                                            attn_opts(**opt), _stream())
    _chk(rc, "tvts_attn_fwd")


def attn_fwd_len(qkv, kv_len, out, lse2, *, B, heads, S, head_dim=64):
    """FULL attention, keys at positions >= kv_len[b] masked (padding)."""
    lib = _lib.load()
    assert kv_len.dtype == torch.int32 and kv_len.numel() == B
    _chk(_attn_fn(lib, "fwd_len", head_dim)(_p(qkv), _ld(qkv), B, heads, S, _p(kv_len), _p(out), _ld(out), _p(lse2), _stream()),
         "tvts_attn_fwd_len")


def attn_bwd_len(qkv, kv_len, dO, O, lse2, delta, dqkv, *, B, heads, S, head_dim=64):
    """backward of attn_fwd_len into dqkv (zeroed here first: the padded positions' dK / dV rows are not written)."""
    lib = _lib.load()
    zero_(dqkv)
    _chk(_attn_fn(lib, "bwd_len", head_dim)(_p(qkv), _ld(qkv), B, heads, S, _p(kv_len), _p(dO), _ld(dO), _p(O), _ld(O), _p(lse2),
                                            _p(delta), _p(dqkv), _ld(dqkv), _stream()), "tvts_attn_bwd_len")


DROP_SITE_STRIDE = 0x632BE59BD9B4E019  # odd 64-bit constant: site k of a step uses seed_dev[0] + k * this (mod 2^64)


def _site(site: int) -> int:
    v = (site * DROP_SITE_STRIDE) & 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v  # the same bits as a C long


def attn_fwd_len_drop(qkv, kv_len, out, lse2, *, B, heads, S, p, seed, site, head_dim=64):
    """attn_fwd_len with dropout p on the attention probabilities; seed: int64 device tensor [1], site: small int (per call site)"""
    lib = _lib.load()
    assert kv_len.dtype == torch.int32 and kv_len.numel() == B and seed.dtype == torch.int64 and seed.is_cuda
    _chk(_attn_fn(lib, "fwd_len_drop", head_dim)(_p(qkv), _ld(qkv), B, heads, S, _p(kv_len), _p(out), _ld(out), _p(lse2), float(p),
                                                 _p(seed), _site(site), _stream()), "tvts_attn_fwd_len_drop")


def attn_bwd_len_drop(qkv, kv_len, dO, O, lse2, delta, dqkv, *, B, heads, S, p, seed, site, head_dim=64):
    lib = _lib.load()
    zero_(dqkv)
    _chk(_attn_fn(lib, "bwd_len_drop", head_dim)(_p(qkv), _ld(qkv), B, heads, S, _p(kv_len), _p(dO), _ld(dO), _p(O), _ld(O),
                                                 _p(lse2), _p(delta), _p(dqkv), _ld(dqkv), float(p), _p(seed), _site(site),
                                                 _stream()), "tvts_attn_bwd_len_drop")


def dropout_rows(x, *, p, seed, site, residual=None, out=None, out_bf16=None):
    """out = x * mask / (1 - p) (+ residual), fp32 and / or bf16 output; the backward is the same call on the gradient."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and seed.dtype == torch.int64
    _chk(lib.tvts_dropout_rows(_p(x), _ld(x), x.shape[0], x.shape[1], float(p), _p(seed), _site(site), _p(residual),
                               _ld(residual) if residual is not None else 0, _p(out), _ld(out) if out is not None else 0,
                               _p(out_bf16), _ld(out_bf16) if out_bf16 is not None else 0, _stream()), "tvts_dropout_rows")


def attn_fwd_tail(qkv, out, lse2, *, B, heads, S, nq, head_dim=64):
    """FULL attention whose only queries are the last nq tokens of every sequence (rows b*S + S-nq .. of out / lse2 are written)."""
    lib = _lib.load()
    _chk(_attn_fn(lib, "fwd_tail", head_dim)(_p(qkv), _ld(qkv), B, heads, S, nq, _p(out), _ld(out), _p(lse2), _stream()),
         "tvts_attn_fwd_tail")


def attn_bwd_tail(qkv, dO, O, lse2, delta, dqkv, *, B, heads, S, nq, head_dim=64):
    """backward of attn_fwd_tail: dQ into the query rows, dK / dV into every row of dqkv; the dQ third of the other rows is the
    caller's to zero."""
    lib = _lib.load()
    _chk(_attn_fn(lib, "bwd_tail", head_dim)(_p(qkv), _ld(qkv), B, heads, S, nq, _p(dO), _ld(dO), _p(O), _ld(O), _p(lse2), _p(delta),
                                             _p(dqkv), _ld(dqkv), _stream()), "tvts_attn_bwd_tail")


def attn_fwd_rowq(qkv, qpos, out, lse2, *, B, heads, S, head_dim=64):
    """one query per sequence at token qpos[b] (int32 device tensor) attending the keys 0 .. qpos[b]; row b*S + qpos[b] of out / lse2."""
    lib = _lib.load()
    assert qpos.dtype == torch.int32 and qpos.numel() == B
    _chk(_attn_fn(lib, "fwd_rowq", head_dim)(_p(qkv), _ld(qkv), B, heads, S, _p(qpos), _p(out), _ld(out), _p(lse2), _stream()),
         "tvts_attn_fwd_rowq")


def attn_bwd_rowq(qkv, qpos, dO, O, lse2, delta, dqkv, *, B, heads, S, head_dim=64):
    """backward of attn_fwd_rowq into dqkv (zeroed here first: only the query row's dQ and the dK / dV of the keys it sees exist)."""
    lib = _lib.load()
    zero_(dqkv)
    _chk(_attn_fn(lib, "bwd_rowq", head_dim)(_p(qkv), _ld(qkv), B, heads, S, _p(qpos), _p(dO), _ld(dO), _p(O), _ld(O), _p(lse2),
                                             _p(delta), _p(dqkv), _ld(dqkv), _stream()), "tvts_attn_bwd_rowq")


def attn_delta(dO, O, delta, *, rows, heads, head_dim=64):
    lib = _lib.load()
    _chk(_attn_fn(lib, "delta", head_dim)(_p(dO), _ld(dO), _p(O), _ld(O), rows, heads, _p(delta), _stream()), "tvts_attn_delta")


def attn_bwd_dq(mode, qkv, dO, lse2, delta, dqkv, *, B, heads, S, T=0, n=0, causal=False, head_dim=64, **opt):
    lib = _lib.load()
    rc = _attn_fn(lib, "bwd_dq", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, int(causal), _p(dO), _ld(dO), _p(lse2),
                              _p(delta), _p(dqkv), _ld(dqkv), attn_opts(**opt), _stream())
    _chk(rc, "tvts_attn_bwd_dq")


def attn_bwd_dkv(mode, qkv, dO, lse2, delta, dqkv, *, B, heads, S, T=0, n=0, causal=False, cls_acc=None, head_dim=64, **opt):
    lib = _lib.load()
    rc = _attn_fn(lib, "bwd_dkv", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, int(causal), _p(dO), _ld(dO),
                               _p(lse2), _p(delta), _p(dqkv), _ld(dqkv), _p(cls_acc), attn_opts(**opt), _stream())
    _chk(rc, "tvts_attn_bwd_dkv")


def attn_fwd_divided(mode, qkv, out, lse2, cls_ws, *, B, heads, S, T, n, head_dim=64, q8out=None, q8_scale=None, q8_amax=None, **opt):
    """Forward of one divided-attention site (patch rows + CLS row); cls_ws is fp32 scratch.  q8out (uint8, out's shape and row
    stride) + q8_scale / q8_amax: the kernels also write the per-tensor e4m3 copy of the output."""
    lib = _lib.load()
    M = B * S
    with _hbm("attn_fwd_" + mode, _nb((qkv, M), (out, M), (q8out, M)) + 8 * M * heads):  # (launches its CLS merge kernel as well)
        if q8out is not None:
            rc = _attn_fn(lib, "fwd_divided_q8", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, _p(out), _ld(out), _p(lse2),
                                                           _p(cls_ws), cls_ws.numel(), _p(q8out), q8out.stride(0), _p(q8_scale),
                                                           _p(q8_amax), attn_opts(**opt), _stream())
        else:
            rc = _attn_fn(lib, "fwd_divided", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, _p(out), _ld(out), _p(lse2),
                                                        _p(cls_ws), cls_ws.numel(), attn_opts(**opt), _stream())
    _chk(rc, "tvts_attn_fwd_divided")


def attn_bwd(mode, qkv, dO, O, lse2, delta, dqkv, *, B, heads, S, T=0, n=0, causal=False, cls_acc=None, head_dim=64, q8out=None,
             q8_scale=None, q8_amax=None, **opt):
    """Whole backward of one attention site into dqkv (delta / cls_acc are scratch).  q8out (uint8, dqkv's shape and row stride): the
    kernels also write the per-tensor e4m3 copy of dqkv (fused divided geometries)."""
    lib = _lib.load()
    M = B * S
    if q8out is not None:
        with _hbm("attn_bwd_" + mode, _nb((qkv, M), (dO, M), (O, M), (dqkv, M), (q8out, M)) + 8 * M * heads):
            rc = _attn_fn(lib, "bwd_q8", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, int(causal), _p(dO), _ld(dO), _p(O),
                                                   _ld(O), _p(lse2), _p(delta), _p(dqkv), _ld(dqkv), _p(cls_acc),
                                                   cls_acc.numel() if cls_acc is not None else 0, _p(q8out), q8out.stride(0),
                                                   _p(q8_scale), _p(q8_amax), attn_opts(**opt), _stream())
        _chk(rc, "tvts_attn_bwd_q8")
        return
    with _hbm("attn_bwd_" + mode, _nb((qkv, M), (dO, M), (O, M), (dqkv, M)) + 8 * M * heads):  # (delta / CLS finalize kernels included)
        rc = _attn_fn(lib, "bwd", head_dim)(MODE[mode], _p(qkv), _ld(qkv), B, heads, S, T, n, int(causal), _p(dO), _ld(dO), _p(O),
                                            _ld(O), _p(lse2), _p(delta), _p(dqkv), _ld(dqkv), _p(cls_acc),
                                            cls_acc.numel() if cls_acc is not None else 0, attn_opts(**opt), _stream())
    _chk(rc, "tvts_attn_bwd")


def attn_cls_finalize(cls_acc, dqkv, *, B, heads, S, head_dim=64):
    lib = _lib.load()
    _chk(_attn_fn(lib, "cls_finalize", head_dim)(_p(cls_acc), B, heads, S, _p(dqkv), _ld(dqkv), _stream()), "tvts_attn_cls_finalize")


def patch_gather(video, keep, out, *, B, T, n, img, patch):
    lib = _lib.load()
    assert video.dtype == torch.float32 and keep.dtype == torch.int32 and video.is_contiguous()
    _chk(lib.tvts_patch_gather(_p(video), _p(keep), B, T, n, img, patch, _p(out), _ld(out), _stream()), "tvts_patch_gather")


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # video_transforms/videoaug.py:17,26


def patch_gather_u8(frames, keep, out, *, B, T, n, img, patch, crop=None, mean=IMAGENET_MEAN, std=IMAGENET_STD, resize=None):
    """frames: uint8 [B, T, H0, W0, 3] on the device; crop: int32 [B, 2] (top, left) or None (centre crop).
    resize = (ytab int32 [H0], xtab int32 [W0]) device tables: the frames are the decoder's pictures and the crop applies to
    their nearest-neighbour resize to H0 x W0 (the reference's video_transform.Resize)."""
    import ctypes
    lib = _lib.load()
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[-1] == 3 and keep.dtype == torch.int32
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    if resize is not None:
        ytab, xtab = resize
        assert ytab.dtype == torch.int32 and xtab.dtype == torch.int32
        _chk(lib.tvts_patch_gather_u8_resized(_p(frames), frames.shape[2], frames.shape[3], _p(ytab), _p(xtab), ytab.numel(),
                                              xtab.numel(), _p(crop), _p(keep), B, T, n, img, patch, m3, s3, _p(out), _ld(out),
                                              _stream()), "tvts_patch_gather_u8_resized")
        return
    _chk(lib.tvts_patch_gather_u8(_p(frames), frames.shape[2], frames.shape[3], _p(crop), _p(keep), B, T, n, img, patch, m3, s3,
                                  _p(out), _ld(out), _stream()), "tvts_patch_gather_u8")


def tube_mask(seed, first_sample, B, ppf, n_keep, out=None, device=None):
    """keep_ind int32 [B, n_keep] drawn on the device: sample number first_sample + b gets the n_keep patch indices with the
    smallest counter-based random keys (an unsorted prefix of a random permutation, YTTemporal_dataset.py:207-213)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(B, n_keep, dtype=torch.int32, device=device if device is not None else "cuda")
    assert out.dtype == torch.int32 and out.is_contiguous() and tuple(out.shape) == (B, n_keep)
    s64 = lambda v: (int(v) & (2 ** 64 - 1)) - (2 ** 64 if int(v) & (1 << 63) else 0)  # unsigned 64-bit through a C long
    _chk(lib.tvts_tube_mask(s64(seed), s64(first_sample), B, ppf, n_keep, _p(out), _stream()), "tvts_tube_mask")
    return out


def vit_assemble(patch, cls, pos, temporal, keep, tok, *, B, T, n):
    """keep [B, n] (one tube mask per clip, v2) or [B, T, n] (one per frame / tubelet, v1)"""
    lib = _lib.load()
    _chk(lib.tvts_vit_assemble(_p(patch), _ld(patch), _p(cls), _p(pos), _p(temporal), _p(keep), 1 if keep.dim() == 3 else 0,
                               B, T, n, tok.shape[1], _p(tok), _ld(tok), _stream()), "tvts_vit_assemble")


def vit_assemble_bwd(dtok, keep, dpatch, dcls, dpos, dtemporal, *, B, T, n):
    lib = _lib.load()
    ws = _tn_workspace(dtok.device)  # ordered partials of the temporal / class embedding sums (shared fp32 scratch of this stream)
    _chk(lib.tvts_vit_assemble_bwd(_p(dtok), _ld(dtok), _p(keep), 1 if keep.dim() == 3 else 0, B, T, n, dtok.shape[1],
                                   _p(dpatch), _ld(dpatch), _p(dcls), _p(dpos), dpos.shape[0] - 1, _p(dtemporal), _p(ws), ws.numel(),
                                   _stream()),
         "tvts_vit_assemble_bwd")


def patch_gather_tube(video, keep, out, *, B, tubes, tubelet, n, img, patch):
    """v1 tubelet im2col: video fp32 [B, T, 3, img, img], keep int32 [B, tubes, n] -> out bf16 [B*tubes*n, 3*tubelet*patch^2]"""
    lib = _lib.load()
    assert video.dtype == torch.float32 and keep.dtype == torch.int32 and keep.dim() == 3
    _chk(lib.tvts_patch_gather_tube(_p(video), _p(keep), B, tubes, tubelet, n, img, patch, _p(out), _ld(out), _stream()),
         "tvts_patch_gather_tube")


def text_embed(ids, emb, pos, x, *, N, L):
    lib = _lib.load()
    assert ids.dtype == torch.int32
    _chk(lib.tvts_text_embed(_p(ids), ids.stride(0), N, L, _p(emb), _p(pos), emb.shape[1], _p(x), _ld(x), _stream()),
         "tvts_text_embed")


def token_sort(ids_cpu):
    """(order, seg) of tvts_text_embed_bwd for the [N, L] token ids of a batch (host tensors in, int32 host tensors out): the rows
    grouped into runs of equal token id (rows of a run in row order) and the starts of the runs padded to N * L + 1 entries.  The
    runs of more than 64 rows come first (the kernel gives each of the first 64 runs a block per 64 columns), the others by id.
    numpy on the host: ~1 ms for the 24 576 tokens of a 192-pair batch (it runs once per step in the trainer's prepare_batch)."""
    import numpy as np
    flat = ids_cpu.reshape(-1).numpy().astype(np.int64, copy=False)
    n = flat.size
    order = np.argsort(flat, kind="stable")
    srt = flat[order]
    first = np.empty(n, dtype=bool)
    first[0] = True
    np.not_equal(srt[1:], srt[:-1], out=first[1:])
    starts = np.flatnonzero(first)
    lens = np.diff(np.append(starts, n))
    if (lens > 64).any():  # long runs to the front: stable, so the others stay in id order
        run_of = np.cumsum(first) - 1                                  # run index of every sorted row
        order = order[np.argsort(lens[run_of] <= 64, kind="stable")]   # rows of long runs first
        lens = lens[np.argsort(lens <= 64, kind="stable")]
        starts = np.cumsum(lens) - lens
    seg = np.full(n + 1, n, dtype=np.int32)
    seg[:starts.size] = starts
    return torch.from_numpy(order.astype(np.int32)), torch.from_numpy(seg)


def text_embed_bwd(dx, ids, demb, dpos, *, N, L, tok_sort=None):
    """tok_sort = (order, seg) device tensors of token_sort(): ordered sums; None: fp32 atomics."""
    lib = _lib.load()
    order, seg = tok_sort if tok_sort is not None else (None, None)
    if order is not None:
        assert order.dtype == torch.int32 and seg.dtype == torch.int32 and order.numel() == N * L and seg.numel() == N * L + 1
    _chk(lib.tvts_text_embed_bwd(_p(dx), _ld(dx), _p(ids), ids.stride(0), N, L, dx.shape[1], _p(demb), _p(dpos),
                                 _p(order), _p(seg), _stream()), "tvts_text_embed_bwd")


def text_mean(t, mean, before, *, NT, B):
    lib = _lib.load()
    _chk(lib.tvts_text_mean(_p(t), NT, B, t.shape[1], _p(mean), _p(before), _stream()), "tvts_text_mean")


def text_mean_bwd(dmean, dt, *, NT, B):
    lib = _lib.load()
    _chk(lib.tvts_text_mean_bwd(_p(dmean), NT, B, dmean.shape[1], _p(dt), _stream()), "tvts_text_mean_bwd")


def sort_assemble(tok, text, type_embed, xs, *, B, S, off, Sv, NT):
    lib = _lib.load()
    _chk(lib.tvts_sort_assemble(_p(tok), _ld(tok), B, S, off, Sv, _p(text), NT, _p(type_embed), xs.shape[1], _p(xs),
                                _ld(xs), _stream()), "tvts_sort_assemble")


def sort_assemble_bwd(dxs, dvid, dout, dtype, *, B, S, off, Sv, NT):
    lib = _lib.load()
    E = dout.shape[1]
    ws = _tn_workspace(dout.device)  # ordered partials of the type-embedding gradient
    _chk(lib.tvts_sort_assemble_bwd(_p(dxs), _ld(dxs) if dxs is not None else 0, B, S, off, Sv, NT, _p(dvid), E,
                                    _p(dout), _ld(dout), _p(dtype), _p(ws), ws.numel(), _stream()), "tvts_sort_assemble_bwd")


def relu(x, out, dy=None):
    """out = relu(x), or with dy: out = dy * (x > 0)"""
    lib = _lib.load()
    _chk(lib.tvts_relu(_p(x), _p(dy), _p(out), x.numel(), _stream()), "tvts_relu")


def rows_gather(src, rows, dst, *, scatter_add=False):
    lib = _lib.load()
    _chk(lib.tvts_rows_gather(_p(src), _ld(src), _p(rows), rows.numel(), src.shape[1], _p(dst), _ld(dst),
                              int(scatter_add), _stream()), "tvts_rows_gather")


def rows_move(mode, rows, *, full_f32=None, full_bf16=None, packed_f32=None, packed_bf16=None):
    """mode "gather": packed[r] = full[rows[r]] (from full_f32 if given, else full_bf16); "scatter": full[rows[r]] = packed[r];
    "scatter_add": full_f32[rows[r]] += packed_f32[r], full_bf16[rows[r]] = bf16(sum).  rows int32, distinct."""
    lib = _lib.load()
    assert rows.dtype == torch.int32
    for t, dt in ((full_f32, torch.float32), (packed_f32, torch.float32), (full_bf16, torch.bfloat16), (packed_bf16, torch.bfloat16)):
        assert t is None or (t.dtype == dt and t.stride(1) == 1)
    ref = packed_f32 if packed_f32 is not None else packed_bf16
    W = ref.shape[1]
    ld = lambda t: _ld(t) if t is not None else 0  # noqa: E731
    _chk(lib.tvts_rows_move({"gather": 0, "scatter": 1, "scatter_add": 2}[mode], _p(rows), rows.numel(), W, _p(full_f32), ld(full_f32),
                            _p(full_bf16), ld(full_bf16), _p(packed_f32), ld(packed_f32), _p(packed_bf16), ld(packed_bf16), _stream()),
         "tvts_rows_move")


def zero_cols_bf16(x, cols):
    """x[:, :cols] = 0 (bf16, row-major)"""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    _chk(lib.tvts_zero_cols_bf16(_p(x), _ld(x), x.shape[0], int(cols), _stream()), "tvts_zero_cols_bf16")


def l2norm_rows(x, xn, inv, eps=1e-8):
    lib = _lib.load()
    _chk(lib.tvts_l2norm_rows(_p(x), x.shape[0], x.shape[1], eps, _p(xn), _p(inv), _stream()), "tvts_l2norm_rows")


def l2norm_rows_bwd(dxn, xn, inv, dx):
    lib = _lib.load()
    _chk(lib.tvts_l2norm_rows_bwd(_p(dxn), _p(xn), _p(inv), xn.shape[0], xn.shape[1], _p(dx), _stream()),
         "tvts_l2norm_rows_bwd")


def infonce(x, lse, dx, loss):
    lib = _lib.load()
    _chk(lib.tvts_infonce(_p(x), x.shape[0], _p(lse), _p(dx), _p(loss), _stream()), "tvts_infonce")


def cross_entropy(logits, labels, scale, dlogits, loss):
    lib = _lib.load()
    assert labels.dtype == torch.int32
    _chk(lib.tvts_cross_entropy(_p(logits), _p(labels), logits.shape[0], logits.shape[1], scale, _p(dlogits), _p(loss),
                                _stream()), "tvts_cross_entropy")


def retrieval_ranks(sims, mode, valid=None):
    """ranks of the ground truth per query of a text x video similarity matrix; mode 't2v' | 'v2t';
    valid: optional uint8 [n_text], 0 where a caption is missing (the reference's query_masks)."""
    lib = _lib.load()
    assert sims.dtype == torch.float32 and sims.dim() == 2 and sims.stride(1) == 1
    nt, nv = sims.shape
    ranks = torch.empty(nt if mode == "t2v" else nv, dtype=torch.float32, device=sims.device)
    assert valid is None or (valid.dtype == torch.uint8 and valid.numel() == nt and valid.is_contiguous())
    _chk(lib.tvts_retrieval_ranks(_p(sims), sims.stride(0), nt, nv, 0 if mode == "t2v" else 1, _p(valid), _p(ranks), _stream()),
         "tvts_retrieval_ranks")
    return ranks


def adamw_hf(p, g, m, v, shadow, chunk_group, lr4, wd4, step, beta1=0.9, beta2=0.999, eps=1e-6, grad_scale=1.0,
             step_dev=None, hyper_dev=None):
    lib = _lib.load()
    lr = (ctypes.c_float * 4)(*lr4)
    wd = (ctypes.c_float * 4)(*wd4)
    # 30 B per element of every chunk that steps: p, m, v read + written (24), g read (4), bf16 shadow written (2)
    with _hbm("adamw", lambda: 30.0 * 1024 * float((chunk_group >= 0).sum()) if chunk_group.dtype != torch.uint8
              else 30.0 * 1024 * float((chunk_group < 255).sum())):
        rc = (lib.tvts_adamw_hf(_p(p), _p(g), _p(m), _p(v), _p(shadow), _p(chunk_group), chunk_group.numel(),
                           ctypes.cast(lr, ctypes.c_void_p), ctypes.cast(wd, ctypes.c_void_p), step, _p(step_dev), _p(hyper_dev), beta1, beta2, eps,
                           grad_scale, _stream()))
    _chk(rc, "tvts_adamw_hf")


def cast_f32_bf16(src, dst):
    lib = _lib.load()
    _chk(lib.tvts_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "tvts_cast_f32_bf16")


def cast_bf16_f32(src, dst):
    lib = _lib.load()
    _chk(lib.tvts_cast_bf16_f32(_p(src), _p(dst), src.numel(), _stream()), "tvts_cast_bf16_f32")


def pad_rows_bf16(src, dst):
    """dst[r, :] = src[r, :] zero-extended to dst's width (bf16, row-major 2-D)."""
    lib = _lib.load()
    _chk(lib.tvts_pad_rows_bf16(_p(src), _ld(src), _p(dst), _ld(dst), src.shape[0], src.shape[1], dst.shape[1], _stream()),
         "tvts_pad_rows_bf16")


def add_rows_f32(dst, src):
    """dst[r, c] += src[r, c] for c < dst.shape[1] (fp32; src may be wider)."""
    lib = _lib.load()
    _chk(lib.tvts_add_rows_f32(_p(dst), _ld(dst), _p(src), _ld(src), dst.shape[0], dst.shape[1], _stream()), "tvts_add_rows_f32")


def transpose_batched(src, dst, tiles, ntiles):
    lib = _lib.load()
    with _hbm("weight_transpose", 4.0 * 64 * 64 * ntiles):
        rc = lib.tvts_transpose_bf16_batched(_p(src), _p(dst), _p(tiles), ntiles, _stream())
    _chk(rc, "tvts_transpose_bf16_batched")


def probe_tr16(inp, out):
    lib = _lib.load()
    _chk(lib.tvts_probe_tr16(_p(inp), _p(out), _stream()), "tvts_probe_tr16")


CLOCK_PROBE = None  # bench.py: dict(buf=int64 [n, 4] device tensor, i=0) while the instrumented step runs (see gemm_nt)


def zero_(t):
    """t[...] = 0 through the library (hipMemsetAsync on the current stream); t contiguous"""
    assert t.is_contiguous()
    _chk(_lib.load().tvts_zero_bytes(_p(t), t.numel() * t.element_size(), _stream()), "tvts_zero_bytes")
    return t


def device_clock_info(device=0):
    """-> (rate of the constant counter in kHz, CU count, the sheet's maximum shader clock in kHz)"""
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _chk(_lib.load().tvts_device_clock_info(int(device), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "tvts_device_clock_info")
    return a.value, b.value, c.value


class Event:
    """HIP event on the launch stream (torch.cuda.Event would do too; this one goes through the C ABI)."""

    def __init__(self):
        self._e = ctypes.c_void_p()
        _chk(_lib.load().tvts_event_create(ctypes.byref(self._e)), "tvts_event_create")

    def record(self):
        _chk(_lib.load().tvts_event_record(self._e, _stream()), "tvts_event_record")

    def elapsed_ms(self, end: "Event") -> float:
        ms = ctypes.c_float()
        _chk(_lib.load().tvts_event_elapsed_ms(self._e, end._e, ctypes.byref(ms)), "tvts_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            _lib.load().tvts_event_destroy(self._e)
        except Exception:
            pass
