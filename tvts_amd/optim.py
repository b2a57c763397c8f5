"""Fused multi-tensor AdamW with Hugging Face semantics over the flat parameter store.

Replaces ``transformers.AdamW(params=optimizer_grouped_parameters)`` built at
v2/train_dist_TVTSv2_ViT_B_16.py:118-125 (defaults betas (0.9, 0.999), eps 1e-6, correct_bias=True).
It is a ``torch.optim.Optimizer`` so param_groups / state_dict keep the reference checkpoint layout
(``state[p] = {step, exp_avg, exp_avg_sq}``), but ``step()`` is ONE kernel launch over the flat buffers.

Gradient handling follows the reference's pinned torch 1.11 ``zero_grad()`` (grads zeroed, not set to
None): every trainable tensor is updated every step, tensors that received no gradient see g = 0.
"""
from __future__ import annotations

import torch

from . import hip as K
from .engine import CH, ParamStore


class FusedHFAdamW(torch.optim.Optimizer):
    def __init__(self, params, store: ParamStore, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0,
                 correct_bias=True, model=None):
        if not correct_bias:
            raise NotImplementedError("correct_bias=False is not built")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)
        if len(self.param_groups) > 4:
            raise ValueError("the fused kernel supports up to 4 parameter groups (the reference uses 4)")
        self.store, self.model = store, model
        dev = store.device
        if store.m is None:
            store.m = torch.zeros_like(store.flat)
            store.v = torch.zeros_like(store.flat)
        ptr2name = {store.p(n).data_ptr(): n for n in store.shapes}
        table = torch.full((store.total // CH,), 255, dtype=torch.uint8)
        self._names = []
        for gi, group in enumerate(self.param_groups):
            names = []
            for p in group["params"]:
                name = ptr2name.get(p.data_ptr())
                if name is None:
                    raise ValueError("FusedHFAdamW only handles parameters of the flat store")
                names.append(name)
                o, n = store.off[name], p.numel()
                table[o // CH:(o + n + CH - 1) // CH] = gi
                o2 = store.off[name]
                self.state[p] = dict(step=0, exp_avg=store.m[o2:o2 + n].view(p.shape), exp_avg_sq=store.v[o2:o2 + n].view(p.shape))
            self._names.append(names)
        self.chunk_group = table.to(dev)
        self.global_step = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)  # lr[4] | wd[4] for captured launches
        self._hyper_host = None
        self.grad_scale = 1.0
        self._ranged = None      # (hyper-parameters, device_step, [chunk ranges done]) between begin_ranges() and step()
        self._opt_stream = None

    def _hyper(self):
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for g in self.param_groups[1:]:  # one kernel launch covers every group: these three cannot differ per group
            if tuple(g["betas"]) != (b1, b2) or g["eps"] != eps:
                raise ValueError("FusedHFAdamW needs the same betas / eps in every parameter group (lr and weight_decay may differ)")
        n = len(self.param_groups)
        lr4 = [float(g["lr"]) for g in self.param_groups] + [0.0] * (4 - n)
        wd4 = [float(g["weight_decay"]) for g in self.param_groups] + [0.0] * (4 - n)
        return b1, b2, eps, lr4, wd4

    def sync_hyper(self):
        """Upload lr / weight_decay of the parameter groups to the device table read by captured (device_step) launches.
        step() calls it; when a captured hipGraph is REPLAYED, call it after changing param_groups (an LR schedule) -- the
        replay then uses the new values without re-capture.  The copy is skipped when nothing changed."""
        *_, lr4, wd4 = self._hyper()
        if self._hyper_host != (lr4, wd4):
            self.hyper_dev.copy_(torch.tensor(lr4 + wd4, dtype=torch.float32), non_blocking=False)
            self._hyper_host = (lr4, wd4)

    def _sync_step_from_device(self):
        """Under hipGraph replay the host counters only advance at capture time: re-read the device counter."""
        st = int(self.step_dev.item())
        if st > self.global_step:
            self.global_step = st
        for s in self.state.values():
            s["step"] = self.global_step

    def state_dict(self):
        self._sync_step_from_device()
        return super().state_dict()

    def zero_grad(self, set_to_none: bool = False):
        K.zero_(self.store.grad)

    def _begin(self, device_step: bool):
        """counters of a new optimizer step (once per step, in front of its first launch)"""
        b1, b2, eps, lr4, wd4 = self._hyper()
        capturing = torch.cuda.is_current_stream_capturing()
        if device_step and not capturing:
            self.sync_hyper()  # (a host -> device copy: not capturable; the caller syncs before capture / replay)
        elif device_step and self._hyper_host != (lr4, wd4):
            raise RuntimeError("FusedHFAdamW: call sync_hyper() (or run one eager device_step) before capturing the step")
        self.global_step += 1
        if device_step:
            self.step_dev.add_(1)
        else:
            self.step_dev.fill_(self.global_step)  # keeps the device counter in step when eager and captured steps mix
        return b1, b2, eps, lr4, wd4

    def _launch(self, lo: int, hi: int, hp, device_step: bool):
        """the update of the chunks [lo, hi) of the flat buffers + the transposed shadows of the weights inside them"""
        b1, b2, eps, lr4, wd4 = hp
        st = self.store
        s, e = lo * CH, hi * CH
        K.adamw_hf(st.flat[s:e], st.grad[s:e], st.m[s:e], st.v[s:e], st.shadow[s:e], self.chunk_group[lo:hi],
                   lr4, wd4, self.global_step, b1, b2, eps, self.grad_scale, step_dev=self.step_dev if device_step else None,
                   hyper_dev=self.hyper_dev if device_step else None)
        st.refresh_transposed(s, e)

    @torch.no_grad()
    def step(self, closure=None, device_step: bool = False):
        """device_step=True keeps the step counter in device memory (hipGraph replay).  After begin_ranges() / step_range() calls
        this finishes the step: the chunks no range has covered yet."""
        nch = self.chunk_group.numel()
        if self._ranged is None:
            hp = self._begin(device_step)
            K.adamw_hf(self.store.flat, self.store.grad, self.store.m, self.store.v, self.store.shadow, self.chunk_group,
                       hp[3], hp[4], self.global_step, hp[0], hp[1], hp[2], self.grad_scale, step_dev=self.step_dev if device_step else None,
                       hyper_dev=self.hyper_dev if device_step else None)
            self.store.refresh_shadows(cast=False)
        else:
            hp, dstep, done = self._ranged
            self._ranged = None
            cur = torch.cuda.current_stream(self.store.device)
            if self._opt_stream is not None and done:
                cur.wait_stream(self._opt_stream)  # every range launched beside the backward has landed
            pos = 0
            for lo, hi in sorted(done) + [(nch, nch)]:
                if lo > pos:
                    self._launch(pos, lo, hp, dstep)
                pos = max(pos, hi)
            self.store.refresh_shadows(cast=False, transposed=False)  # what is not per-range: K-padded conv weight, e4m3 copies
        if self.model is not None:
            self.model.mark_shadows_fresh()
        if not device_step:
            for st in self.state.values():
                st["step"] = self.global_step

    # ---- the update range by range, beside the backward (round 5).  The hand-written backward reports every parameter range whose
    # gradient is final (Engine._ready -> param_ready); its update -- HBM-bound, 30 B per parameter -- then runs on a stream of its
    # own under the remaining backward instead of behind it: the layer's weights are not read again before the next forward.
    # begin_ranges() once per step in front of the backward, step_range(lo, hi) per finished range (elements of the flat buffers,
    # chunk-aligned as Engine.trainable_runs hands them out), step() afterwards for whatever no range covered.
    def begin_ranges(self, device_step: bool = False):
        self._ranged = (self._begin(device_step), device_step, [])

    def step_range(self, lo_elem: int, hi_elem: int, wait=None):
        """wait: callable run on the optimizer stream in front of the launch (the range's all-reduce completing)"""
        if self._ranged is None:
            raise RuntimeError("FusedHFAdamW.step_range outside begin_ranges() ... step()")
        hp, dstep, done = self._ranged
        lo, hi = lo_elem // CH, -(-hi_elem // CH)
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.store.device)
        self._opt_stream.wait_stream(torch.cuda.current_stream(self.store.device))
        with torch.cuda.stream(self._opt_stream), K.lane(2):
            if wait is not None:
                wait()
            self._launch(lo, hi, hp, dstep)
        done.append((lo, hi))

    def load_state_dict(self, state_dict):
        """Accepts the HF-AdamW / torch layout: state keyed by the running parameter index."""
        sd_groups, sd_state = state_dict["param_groups"], state_dict["state"]
        for g, sg in zip(self.param_groups, sd_groups):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            for p, idx in zip(g["params"], sg["params"]):
                st = sd_state.get(idx)
                if st is None:
                    continue
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                self.state[p]["step"] = int(st["step"])
                self.global_step = max(self.global_step, int(st["step"]))
        self.step_dev.fill_(self.global_step)
