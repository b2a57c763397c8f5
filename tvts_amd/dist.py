"""Data-parallel exchange steps of the TVTSv2 step: one process per GPU.

* ``allgather_embeds`` / ``EmbedGather`` -- the reference's ``AllGather_multi`` (v2/trainer/trainer.py:41-57): forward
  gathers the per-rank ``[B,E]`` embeddings to ``[W*B,E]``; backward is a LOCAL ROW SLICE of the incoming gradient, with
  no collective (``local_rows``).  Video and text embeddings travel in one message.
* ``GradSync`` -- the DDP gradient average (v2/base/base_trainer.py:20-25) over the flat fp32 gradient buffer: ranges
  are all-reduced (SUM) asynchronously as soon as the hand-written backward has finished them, overlapping RCCL traffic
  over xGMI with the remaining backward GEMMs; the 1/W factor is folded into the AdamW kernel.  Ranges of frozen
  parameters never reach it (Engine._ready hands over trainable runs only).  Payload fp32 (default, bit-faithful to the
  reference's fp32 all-reduce) or bf16 (half the bytes on the links: 312 MB instead of 624 MB for ViT-B/16;
  ``TVTS_GRAD_PAYLOAD=bf16``).

Two transports, same semantics:
  ``native`` the C ABI of include/tvts_comm.h: RCCL calls on a side HIP stream the library owns, fork / join by events against
             the compute stream, no host synchronisation anywhere in the step.  OPT-IN (``TVTS_COMM=native``) since round 4:
             never run on more than one GPU, and its bring-up can hang instead of falling back (see transport());
  ``torch``  torch.distributed collectives (backend nccl = RCCL on the GPUs, gloo on CPU tensors in the tests): the default.
``TVTS_COMM`` is read ONCE, by the first call of transport(); the result is cached for the process and later changes of the
environment variable are ignored (every rank has to stay on the transport the ranks agreed on).

CU reservation (``TVTS_NT_CUS``, default none): the persistent 256x256 GEMM blocks fill a CU completely, so an RCCL kernel only
gets CUs at a GEMM kernel boundary or on CUs the persistent grid leaves free (``TVTS_NT_CUS=auto`` applies the rule of
auto_cu_reservation, an unmeasured prediction, hence opt-in).  tools/overlap_probe.py (profiles/r03_overlap_probe.txt)
measures both on one GPU: a side-stream kernel forked under the dgrad chain starts within one GEMM launch (<= 0.5 ms) either way;
on 8 reserved CUs it streams at ~0.27 TB/s (128 MB in 480 us), on 32 at ~1.3 TB/s -- and reserving 8 / 16 CUs costs the GEMM chain
4.3 % / 8.7 %.  At the bench's 192 pairs per GPU the whole gradient all-reduce (624 MB fp32) is ~5 ms of a 150 ms step, less
than what a reservation costs, so none is made by default; at small per-GPU batches set TVTS_NT_CUS=248.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def allgather_embeds(video_emb: torch.Tensor, text_emb: torch.Tensor, scratch=None):
    """-> (video_all [W*B,E], text_all [W*B,E]); rank r owns rows r*B..(r+1)*B-1."""
    W, _ = world()
    if W == 1:
        return video_emb, text_emb
    B, E = video_emb.shape
    packed = torch.cat([video_emb, text_emb], dim=1).contiguous()
    out = torch.empty(W * B, 2 * E, dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    return out[:, :E].contiguous(), out[:, E:].contiguous()


def local_rows(grad_all: torch.Tensor, B: int) -> torch.Tensor:
    """AllGather_multi.backward: this rank's rows of the gradient wrt the gathered tensor."""
    _, r = world()
    return grad_all[B * r:B * (r + 1)]


# ------------------------------------------------------------------------------------------------ native transport
class NativeComm:
    """ctypes binding of libtvts_comm.so (include/tvts_comm.h); the unique id travels over the torch.distributed group
    that the launcher initialised anyway (rendezvous is plumbing)."""

    _inst: Optional["NativeComm"] = None

    def __init__(self):
        from . import _lib
        self.lib = _lib.load_comm()
        W, r = world()
        dev = torch.device("cuda", torch.cuda.current_device())
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if r == 0:
            raw = (ctypes.c_ubyte * 128)()
            if self.lib.tvts_comm_unique_id(ctypes.cast(raw, ctypes.c_void_p)) == 0:
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
        if W > 1:  # the broadcast runs on every rank even when rank 0 has no id to send (all zero), so that all of them fail together
            t = idbuf.to(dev) if dist.get_backend() == "nccl" else idbuf
            dist.broadcast(t, src=0)
            idbuf = t.cpu()
        if not bool(idbuf.any()):
            raise RuntimeError("tvts_comm_unique_id failed on rank 0")
        raw = (ctypes.c_ubyte * 128)(*idbuf.tolist())
        h = ctypes.c_void_p()
        # the rendezvous has a deadline (TVTS_COMM_TIMEOUT_S, default 60 s): a rank that died after the id broadcast costs the others
        # the deadline, then they are back here with an error and reach transport()'s agreement instead of hanging in RCCL
        self.timeout_s = float(os.environ.get("TVTS_COMM_TIMEOUT_S", "60"))
        self._chk(self.lib.tvts_comm_create_deadline(ctypes.cast(raw, ctypes.c_void_p), r, W, int(self.timeout_s * 1000), ctypes.byref(h)),
                  "tvts_comm_create_deadline")
        self.h, self.W, self.rank = h, W, r
        import atexit
        atexit.register(self.close)  # before torch.distributed tears its own RCCL communicators down at interpreter exit

    def abort(self):
        """Tears the communicator down without waiting for its outstanding collectives (a peer is gone): ncclCommAbort."""
        h, self.h = self.h, None
        if h is not None:
            self.lib.tvts_comm_abort(h)
            if NativeComm._inst is self:
                NativeComm._inst = None

    def drain(self, timeout_s: float) -> bool:
        """Polls (never blocks inside the driver) until every collective issued so far has completed; False after timeout_s."""
        import time
        t0 = time.time()
        while True:
            st = self.lib.tvts_comm_idle(self.h)
            if st == 1:
                return True
            if st < 0:
                raise RuntimeError(f"tvts_comm_idle failed with code {st}")
            # timeout_s <= 0 means "no deadline", as it does on the C side (tvts_comm_create_deadline falls through to a blocking
            # create): poll until done instead of giving up at the first poll and aborting a healthy communicator
            if timeout_s > 0 and time.time() - t0 > timeout_s:
                return False
            time.sleep(0.002)

    def close(self):
        """Destroys the communicator, the side stream and its events (idempotent)."""
        h, self.h = self.h, None
        if h is not None:
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self.lib.tvts_comm_destroy(h)
            if NativeComm._inst is self:
                NativeComm._inst = None

    @staticmethod
    def _chk(rc, name):
        if rc != 0:
            raise RuntimeError(f"{name} failed with code {rc}")

    @classmethod
    def get(cls) -> "NativeComm":
        if cls._inst is None:
            cls._inst = NativeComm()
        return cls._inst

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def allgather_embeds(self, video, text, video_all, text_all):
        B, E = video.shape
        self._chk(self.lib.tvts_comm_allgather_embeds(self.h, ctypes.c_void_p(video.data_ptr()), ctypes.c_void_p(text.data_ptr()),
                                                     B, E, ctypes.c_void_p(video_all.data_ptr()),
                                                     ctypes.c_void_p(text_all.data_ptr()), self._stream()),
                  "tvts_comm_allgather_embeds")

    def allreduce(self, t: torch.Tensor):
        dt = {torch.float32: 0, torch.bfloat16: 1}[t.dtype]
        self._chk(self.lib.tvts_comm_allreduce_bucket(self.h, ctypes.c_void_p(t.data_ptr()), t.numel(), dt, self._stream()),
                  "tvts_comm_allreduce_bucket")

    def wait(self):
        self._chk(self.lib.tvts_comm_wait(self.h, self._stream()), "tvts_comm_wait")

    def self_test(self):
        """One small all-reduce and one all-gather with known answers (rank r contributes r + 1); raises when a result is wrong."""
        W, r = self.W, self.rank
        dev = torch.device("cuda", torch.cuda.current_device())
        x = torch.full((4096,), float(r + 1), device=dev)
        v = torch.full((2, 8), float(r + 1), device=dev)
        t = -v
        v_all, t_all = torch.zeros(2 * W, 8, device=dev), torch.zeros(2 * W, 8, device=dev)
        self.allreduce(x)
        self.allgather_embeds(v, t, v_all, t_all)
        # NOT tvts_comm_wait + synchronize: with a dead peer the collectives never complete and the compute stream would be wedged
        # behind them.  The side stream is polled with a deadline; on a timeout the communicator is aborted (ends the kernels).
        if not self.drain(self.timeout_s):
            self.abort()
            raise RuntimeError(f"native transport self-test: collectives did not complete within {self.timeout_s:.0f} s (communicator aborted)")
        self.wait()
        torch.cuda.synchronize()
        want = torch.arange(1, W + 1, device=dev, dtype=torch.float32).repeat_interleave(2)[:, None].expand(2 * W, 8)
        if not (bool((x == W * (W + 1) / 2).all()) and torch.equal(v_all, want) and torch.equal(t_all, -want)):
            raise RuntimeError("native transport self-test: wrong all-reduce / all-gather result")


_TRANSPORT: Optional[str] = None


def auto_cu_reservation(rows_per_gpu: int, world_size: int, grad_bytes: int = 624 << 20) -> int:
    """CUs the persistent GEMM grids would leave to the RCCL kernels (0 = none), from the per-GPU token rows and the world size.
    An UNMEASURED prediction (no multi-GPU box so far): it is applied only when asked for, TVTS_NT_CUS=auto; the default is no
    reservation, a number in TVTS_NT_CUS is the grid size to use.

    Measured on one GPU (tools/overlap_probe.py, profiles/r03_overlap_probe.txt): a side-stream kernel forked under the dgrad chain
    starts within one GEMM launch either way; on 8 reserved CUs it streams 0.27 TB/s, and reserving 8 / 16 CUs costs the GEMM chain
    4.3 % / 8.7 %.  Model: the gradient all-reduce moves 2 (W - 1) / W x grad_bytes per GPU at ~300 GB/s of bus bandwidth (xGMI ring),
    the backward lasts 3.7 ms + 0.62 us per token row; a reservation pays when the traffic it keeps moving is worth more than the
    4.3 % it takes, taken as comm > 7 % of the backward:
      * rows < 15 000 (the reference's 12 pairs): the ViT's GEMMs run on the 128 x 128 kernels (two blocks per CU, 128 of 160 KiB of
        LDS, half the registers) -- RCCL's workgroups co-reside, nothing to reserve;
      * above that every CU is held by a persistent 256 x 256 block: 8 CUs while comm > 0.07 x backward, which with the model's
        numbers is up to ~82 000 rows at W = 8 (3.6 ms of ring time; ~105 pairs of B/16) and up to ~44 000 rows at W = 2 (2.2 ms);
      * beyond: the all-reduce hides at the kernel boundaries -- nothing.
    The decision is taken ONCE, from the first batch's rows (a loop that alternates loaders of different clip lengths keeps it)."""
    if world_size <= 1:
        return 0
    comm_ms = 2.0 * (world_size - 1) / world_size * grad_bytes / 300e9 * 1e3
    bwd_ms = 3.7 + 0.62e-3 * rows_per_gpu
    if rows_per_gpu < 15000:
        return 0
    return 8 if comm_ms > 0.07 * bwd_ms else 0


_NT_CUS_APPLIED = False
CU_RESERVATION = {"nt_cus": None, "source": "none (default: no reservation; TVTS_NT_CUS=<grid> or =auto)"}  # what the bench line reports


def _apply_cu_reservation(rows_per_gpu: Optional[int] = None):
    """The persistent GEMM grids leave CUs to the RCCL kernels ONLY when TVTS_NT_CUS asks for it: a number is the grid size
    (e.g. 248 on the 256-CU part), `auto` applies auto_cu_reservation() once the per-GPU token rows are known (StepRunner's first
    step at world > 1).  Applied once per process."""
    global _NT_CUS_APPLIED
    if _NT_CUS_APPLIED:
        return
    env = os.environ.get("TVTS_NT_CUS")
    if env is None:
        return  # opt-in: nothing is reserved by default (the rule is a prediction until a multi-GPU run has measured it)
    if env == "auto" and rows_per_gpu is None:
        return  # nothing to decide yet
    _NT_CUS_APPLIED = True
    if env == "auto":
        W, _ = world()
        keep = auto_cu_reservation(rows_per_gpu, W)
        n_cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256
        cus = (n_cus - keep) if keep else 0
        CU_RESERVATION.update(nt_cus=cus or None, source=f"TVTS_NT_CUS=auto: {keep} of {n_cus} CUs at {rows_per_gpu} rows per GPU, world {W}")
    else:
        cus = int(env)
        CU_RESERVATION.update(nt_cus=cus or None, source=f"TVTS_NT_CUS={env}")
    if cus:
        from . import hip as K
        K.set_default(nt_cus=cus)


def transport() -> str:
    """'native' or 'torch' (see the module docstring); decided once per process.

    Round 4: the native transport is OPT-IN again (TVTS_COMM=native).  It has still never run on more than one MI355X, and its
    bring-up is not hang-proof: if one rank fails inside communicator creation AFTER the id broadcast (ncclCommInitRank,
    hipStreamCreate), the others are already blocked inside ncclCommInitRank / the self-test's collectives on the side stream
    and never reach the agreement below -- a hang where the documented behaviour is a fallback.  Until it has run on real
    multi-GPU hardware the default at every world size is torch.distributed (RCCL underneath on the nccl backend)."""
    global _TRANSPORT
    if _TRANSPORT is not None:
        return _TRANSPORT
    t = os.environ.get("TVTS_COMM", "torch")
    if t not in ("torch", "native"):
        raise ValueError(f"TVTS_COMM={t!r}: expected 'torch' or 'native'")
    _TRANSPORT = "torch"
    W, _ = world()
    if t == "native" and torch.cuda.is_available() and (W == 1 or dist.get_backend() == "nccl"):
        # Every rank must end up on the same transport: each stage (the library loads; the communicator comes up and a small
        # all-reduce / all-gather with known answers is right) is followed by an agreement over torch.distributed, and one
        # rank's failure at those points sends all of them to the torch transport.
        def agree(ok: bool) -> bool:
            if W == 1:
                return ok
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())
        err = None
        try:
            from . import _lib
            _lib.load_comm()
        except Exception as e:  # the library is optional: torch.distributed carries the same exchange steps
            err = e
        if agree(err is None):
            try:
                NativeComm.get().self_test()
            except Exception as e:
                err = e
            if agree(err is None):
                _TRANSPORT = "native"
            elif NativeComm._inst is not None:
                # another rank failed: this rank's communicator may have collectives in flight towards it -> abort, do not wait
                NativeComm._inst.abort()
        if _TRANSPORT != "native":
            import warnings
            why = f"{type(err).__name__}: {err}" if err is not None else "another rank failed"
            warnings.warn(f"native transport (libtvts_comm.so) not usable ({why}); exchange steps go through torch.distributed")
    if _TRANSPORT == "native":
        _apply_cu_reservation()
    return _TRANSPORT


# ------------------------------------------------------------------------------------------------ embedding all-gather
class EmbedGather:
    """The exchange started early: `start` is called by the engine as soon as both embeddings exist (before the sort
    head's forward), the collective runs asynchronously, `result` waits for it where the loss needs the gathered rows
    (trainer.py:479-483).  Buffers are allocated once per shape."""

    def __init__(self, native: Optional[bool] = None):
        self.W, _ = world()
        self.pending = None
        self.native = (transport() == "native") if native is None else native
        self._buf = {}

    def _bufs(self, B, E, ref):
        key = (B, E, ref.dtype, ref.device)
        b = self._buf.get(key)
        if b is None:
            if self.native:
                b = (torch.empty(self.W * B, E, dtype=ref.dtype, device=ref.device),
                     torch.empty(self.W * B, E, dtype=ref.dtype, device=ref.device))
            else:
                b = (torch.empty(B, 2 * E, dtype=ref.dtype, device=ref.device),
                     torch.empty(self.W * B, 2 * E, dtype=ref.dtype, device=ref.device),
                     torch.empty(self.W * B, E, dtype=ref.dtype, device=ref.device),
                     torch.empty(self.W * B, E, dtype=ref.dtype, device=ref.device))
            self._buf[key] = b
        return b

    def start(self, text_emb: torch.Tensor, video_emb: torch.Tensor):
        if self.W == 1 and not self.native:
            self.pending = ("local", video_emb, text_emb)
            return
        B, E = video_emb.shape
        if self.native:
            v_all, t_all = self._bufs(B, E, video_emb)
            NativeComm.get().allgather_embeds(video_emb, text_emb, v_all, t_all)
            self.pending = ("native", v_all, t_all)
            return
        packed, out, v_all, t_all = self._bufs(B, E, video_emb)
        packed[:, :E].copy_(video_emb)
        packed[:, E:].copy_(text_emb)
        self.pending = ("torch", dist.all_gather_into_tensor(out, packed, async_op=True), out, v_all, t_all, E)

    def result(self):
        p, self.pending = self.pending, None
        if p[0] == "local":
            return p[1], p[2]
        if p[0] == "native":
            NativeComm.get().wait()
            return p[1], p[2]
        _, h, out, v_all, t_all, E = p
        h.wait()
        v_all.copy_(out[:, :E])
        t_all.copy_(out[:, E:])
        return v_all, t_all


# ------------------------------------------------------------------------------------------------ gradient all-reduce
class GradSync:
    """Asynchronous bucketed all-reduce of ranges of one flat gradient buffer."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20, payload: Optional[str] = None,
                 native: Optional[bool] = None):
        self.g = flat_grad
        self.W, _ = world()
        self.handles: List = []
        self.payload = payload or os.environ.get("TVTS_GRAD_PAYLOAD", "fp32")
        if self.payload not in ("fp32", "bf16"):
            raise ValueError(f"gradient payload {self.payload!r}: expected 'fp32' or 'bf16'")
        self.native = ((transport() == "native") if native is None else native) and flat_grad.is_cuda
        esize = 2 if self.payload == "bf16" else flat_grad.element_size()
        self.bucket_elems = max(1, bucket_bytes // esize)
        # bf16 payload: a staging buffer the size of the gradient, filled / read back range by range
        self.stage = (torch.empty(flat_grad.shape, dtype=torch.bfloat16, device=flat_grad.device)
                      if (self.payload == "bf16" and (self.W > 1 or self.native)) else None)
        self.bytes_sent = 0  # per step, for the tests and the bench line

    def _cast(self, src, dst):
        if src.is_cuda:
            from . import hip as K
            (K.cast_f32_bf16 if dst.dtype == torch.bfloat16 else K.cast_bf16_f32)(src, dst)
        else:
            dst.copy_(src)

    def reduce_range(self, start: int, end: int):
        """Called once the backward no longer writes grad[start:end].  Returns None, or a callable that makes the CURRENT stream
        wait for this range's reductions (and casts a bf16 payload back): the optimizer's range-wise update calls it on its own
        stream in front of the range's AdamW launch; ranges nobody waited for are waited for by finish()."""
        if (self.W == 1 and not self.native) or end <= start:
            return None
        first = len(self.handles)
        for s in range(start, end, self.bucket_elems):
            e = min(end, s + self.bucket_elems)
            if self.stage is not None:
                buf = self.stage[s:e]
                self._cast(self.g[s:e], buf)
            else:
                buf = self.g[s:e]
            self.bytes_sent += buf.numel() * buf.element_size()
            if self.native:
                NativeComm.get().allreduce(buf)
                self.handles.append((None, s, e))
            else:
                self.handles.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), s, e))
        mine = self.handles[first:]

        def wait():
            if self.native:
                NativeComm.get().wait()  # (the current stream waits for everything the side stream has been given so far)
            for ent in mine:
                h, s, e = ent
                if ent in self.handles:
                    self.handles.remove(ent)
                    if h is not None:
                        h.wait()
                    if self.stage is not None:
                        self._cast(self.stage[s:e], self.g[s:e])
        return wait

    def finish(self) -> float:
        """Wait for outstanding reductions; returns the scale (1/W) still to be applied to the sum."""
        if self.native and self.handles:
            NativeComm.get().wait()
        for h, s, e in self.handles:
            if h is not None:
                h.wait()
            if self.stage is not None:
                self._cast(self.stage[s:e], self.g[s:e])
        self.handles = []
        return 1.0 / self.W
