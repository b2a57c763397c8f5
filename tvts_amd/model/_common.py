"""nn.Module surface of the reference model classes over the HIP step engine.

Mirrors v2/model/model_dist_TVTSv2_ViT_B_16.py: ``TVTSv2_*(args, load_checkpoint)``, ``forward(data,
return_embeds=True) -> (text_embeds, video_embeds, pred_order)``, the same parameter names / shapes /
registration order (so ``named_parameters()`` drives the kept entrypoint's substring grouping and the
checkpoint optimizer-state indexing unchanged), ``sim_matrix`` as an importable free function.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .. import hip as K
from ..arch import ARCHS, param_shapes
from ..engine import Engine, LossHead, ParamStore


def _require_gpu(device_index: int) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("tvts_amd needs an MI355X (no CPU / eager fallback exists); torch.cuda.is_available() is False")
    from .. import _lib
    _lib.load()
    return torch.device(f"cuda:{device_index}")


def reference_init_(store: ParamStore, seed: int = 0):
    """Random initialisation with the reference's distributions (no pretrained CLIP weights ship here):
    CLIP.initialize_parameters for the text tower (v2/CLIP/clip/model.py:301-328), the VisionTransformer
    ctor for the ViT (video_encoder_ViT_B_16.py:155-167), the 'zeros' timeattn init (:28-34), default
    nn.Linear / nn.Conv2d / nn.LayerNorm inits elsewhere, type_embed = 0 (sort_transformer.py:101)."""
    a = store.arch
    g = torch.Generator().manual_seed(seed)
    Wt, W, Lt = a["text_width"], a["width"], a["text_layers"]
    proj_std, attn_std, fc_std = (Wt ** -0.5) * ((2 * Lt) ** -0.5), Wt ** -0.5, (2 * Wt) ** -0.5
    for name, shape in store.shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        n = int(np.prod(shape))
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]

        def normal(std):
            return torch.randn(shape, generator=g) * std

        def kaiming_uniform():
            bound = 1.0 / math.sqrt(fan_in)
            return (torch.rand(shape, generator=g) * 2 - 1) * bound
        if any(s in name for s in ("ln_", "norm")) and "type_embed" not in name:
            t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif name == "text_token_embedding.weight":
            t = normal(0.02)
        elif name == "text_positional_embedding":
            t = normal(0.01)
        elif name == "text_projection":
            t = normal(Wt ** -0.5)
        elif name.startswith("text_model"):
            if leaf == "in_proj_weight": t = normal(attn_std)
            elif "out_proj.weight" in name or "c_proj.weight" in name: t = normal(proj_std)
            elif "c_fc.weight" in name: t = normal(fc_std)
            elif leaf == "in_proj_bias" or "out_proj.bias" in name: t = torch.zeros(shape)
            else: t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(Wt * (4 if "c_proj" in name else 1))
        elif name in ("video_model.class_embedding", "video_model.positional_embedding", "video_model.proj",
                      "video_model.temporal_embedding"):
            t = normal(W ** -0.5)
        elif "timeattn.qkv" in name or "timeattn.proj.bias" in name:
            t = torch.zeros(shape)
        elif "timeattn.proj.weight" in name:
            t = torch.ones(shape)
        elif name == "pred_model.type_embed":
            t = torch.zeros(shape)
        elif leaf == "bias":
            # nn.Linear bias: U(-1/sqrt(fan_in of the weight)); fan_in = input features
            w_shape = store.shapes[name[:-4] + "weight"]
            bound = 1.0 / math.sqrt(int(np.prod(w_shape[1:])))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            t = kaiming_uniform()
        store.p(name).copy_(t.to(store.device))
    assert n >= 0


class _ModelFn(torch.autograd.Function):
    """One autograd node for the whole model: forward/backward run the hand-written engine; parameter
    gradients are accumulated straight into the flat grad buffer whose views are the parameters' .grad."""

    @staticmethod
    def forward(ctx, anchor, module, pb):
        eng = module.engine
        te, ve, pred = eng.forward(pb)
        ctx.module, ctx.has_pred = module, pred is not None
        outs = (te.clone(), ve.clone(), pred.view(pb["B"], pb["NT"], -1).clone() if pred is not None else te.new_zeros(1))
        return outs

    @staticmethod
    def backward(ctx, d_te, d_ve, d_pred):
        m = ctx.module
        m._sync_requires_grad()
        foreign = m._prepare_grad_buffer()
        dp = d_pred.reshape(-1, d_pred.shape[-1]).contiguous().float() if ctx.has_pred else None
        m.engine.backward(d_te.contiguous().float(), d_ve.contiguous().float(), dp)
        if hasattr(m.engine, "end_step"):
            m.engine.end_step()  # e4m3 weight gradients: without it the autograd path would stay in its calibration step forever
        m._install_grads()
        for p, name in foreign:  # .grad tensors that are not views of the flat buffer: ordinary accumulation
            p.grad.add_(m.store.g(name))
        return None, None, None


class TVTSv2Base(nn.Module):
    ARCH_NAME = None
    ENGINE = Engine
    INIT = staticmethod(reference_init_)

    def __init__(self, args, load_checkpoint=None, arch=None, init_seed=0, pretrained=None):
        """``args`` / ``load_checkpoint``: the reference's constructor (model_dist_TVTSv2_ViT_B_16.py:11-14).

        ``pretrained`` -- what an empty ``load_checkpoint`` initialises from (ignored when a checkpoint is given, whose strict load
        overwrites every tensor anyway):
          None   the reference's behaviour for the named classes (TVTSv2_B_32 / B_16 / H_14): the pretrained CLIP / OpenCLIP model
                 of the kept packages (model/clip_init.py), an error if it cannot be loaded -- never a silent random start;
                 with an explicit ``arch`` dict (tests, benches, smoke) it means False;
          False  the constructor's random initialisation only (reference_init_);
          a mapping / a path: a CLIP-layout state dict (full model: ``visual.*``, ``transformer.*``, ``token_embedding.weight``, ...)."""
        super().__init__()
        self.args = args
        if pretrained is None and (arch is not None or self.ARCH_NAME is None or ARCHS[self.ARCH_NAME].get("family") == "v1"):
            pretrained = False
        if isinstance(pretrained, str) and pretrained == "reference":  # (a subclass that builds its own arch dict asks for the reference's source)
            pretrained = None
        self.arch = dict(arch if arch is not None else ARCHS[self.ARCH_NAME])
        self.num_clips = 4
        self.n_trans = self.arch["n_trans"]
        dev = _require_gpu(getattr(args, "local_rank", 0))
        self.store = ParamStore(self.arch, dev)
        self.engine = self.ENGINE(self.store)
        self.INIT(self.store, init_seed)
        self.initialised_from = "random"
        if load_checkpoint in ["", None] and pretrained is not False:
            from . import clip_init
            if pretrained is None:
                sd = clip_init.load_reference_clip(self.arch)
            elif isinstance(pretrained, (str, bytes)) or hasattr(pretrained, "__fspath__"):  # a path
                sd = clip_init.load_clip_file(pretrained)
            else:
                sd = pretrained
            clip_init.apply_clip_init_(self.store, sd)
            self.initialised_from = "clip"
            print("ViT initialized with {} weights.".format("OpenCLIP" if self.arch.get("block_order") == "openclip" else "CLIP"))
        self._register_tree()
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._versions = None
        if load_checkpoint not in ["", None]:
            ckpt = torch.load(load_checkpoint, map_location=dev, weights_only=False)  # carries the ConfigParser object (base_trainer.py:171)
            sd = ckpt["state_dict"]
            if next(iter(sd)).startswith("module."):  # utils/util.py:25-50 semantics
                sd = {k[7:]: v for k, v in sd.items()}
            self.load_state_dict(sd, strict=True)
            print("loading checkpoint from {}".format(load_checkpoint))

    # parameters live in the flat store; the module tree only carries the reference's names
    def _register_tree(self):
        for name in self.store.shapes:
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(self.store.p(name)))

    def _named(self):
        if not hasattr(self, "_pmap"):
            object.__setattr__(self, "_pmap", dict(self.named_parameters()))
        return self._pmap

    def _sync_requires_grad(self):
        pm = self._named()
        for name in self.store.shapes:
            self.engine.requires_grad[name] = bool(pm[name].requires_grad)

    def _prepare_grad_buffer(self):
        """autograd semantics for the flat gradient buffer the engine accumulates (+=) into: a parameter whose .grad is
        None (optimizer.zero_grad() defaults to set_to_none=True) starts from zero, one whose .grad is the flat view keeps
        accumulating, one carrying a foreign .grad tensor gets this backward's gradient added to it afterwards."""
        pm, st = self._named(), self.store
        none, foreign, kept = [], [], 0
        for name, p in pm.items():
            if not p.requires_grad:
                continue
            if p.grad is None:
                none.append(name)
            elif p.grad.data_ptr() == st.g(name).data_ptr():
                kept += 1
            else:
                none.append(name)
                foreign.append((p, name))
        if kept == 0:
            st.grad.zero_()  # the common case: one launch
        else:
            for name in none:
                st.g(name).zero_()
        return foreign

    def _install_grads(self):
        for name, p in self._named().items():
            if p.requires_grad and p.grad is None:
                p.grad = self.store.g(name)

    def _fresh_shadows(self):
        """Re-derive the bf16 weight shadows iff some parameter changed since the last refresh."""
        vers = tuple(p._version for p in self._named().values())
        if vers != self._versions or self.store.shadow_version < 0:
            for name, p in self._named().items():
                if p.data_ptr() != self.store.p(name).data_ptr():
                    raise RuntimeError(f"parameter {name} no longer aliases the flat store (model moved / cast?)")
            self.store.refresh_shadows()
            self._versions = vers
            self.store.shadow_version = 0

    def mark_shadows_fresh(self):
        self._versions = tuple(p._version for p in self._named().values())
        self.store.shadow_version = 0

    def set_device(self, device):
        self.device = device

    def __str__(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + "\nTrainable parameters: {}".format(n)

    def forward(self, data, return_embeds=True):
        self._fresh_shadows()
        pb = self.engine.prepare_batch(data)
        if torch.is_grad_enabled():
            te, ve, pred = _ModelFn.apply(self._anchor, self, pb)
            if pb["NT"] == 1:
                pred = None
        else:
            te, ve, pred = self.engine.forward(pb)
            te, ve = te.clone(), ve.clone()
            pred = pred.view(pb["B"], pb["NT"], -1).clone() if pred is not None else None
        if return_embeds:
            return te, ve, pred
        return sim_matrix(te, ve)

    # pieces the reference exposes (model_dist_TVTSv2_ViT_B_16.py:97-116)
    def compute_text(self, text_data):
        self._fresh_shadows()
        ids = text_data.detach().to("cpu", torch.int64)
        eot = ids.argmax(-1)
        L, N = int(eot.max()) + 1, ids.shape[0]
        rows = (torch.arange(N) * L + eot).to(torch.int32).to(self.store.device)
        t = self.engine.text_forward(ids[:, :L].to(torch.int32).contiguous().to(self.store.device), rows, N, L).clone()
        return t, t

    def compute_video(self, video_data, keep_ind):
        self._fresh_shadows()
        if video_data.dim() == 4:
            video_data = video_data.unsqueeze(1)
        v = video_data.to(self.store.device).contiguous() if video_data.dtype == torch.uint8 else \
            video_data.to(self.store.device, torch.float32).contiguous()
        self.engine.ctx = dict(self.engine.ctx, crop=None)
        B, T = v.shape[:2]
        keep = keep_ind.to(torch.int32).contiguous().to(self.store.device)
        out, pooled = self.engine.video_forward(v, keep, B, T)
        out = out.view(B, -1, self.arch["embed"]).clone()
        if pooled is None:  # B models (model_dist_TVTSv2_ViT_B_16.py:113-116): all tokens, CLS row is the embedding
            return out, out[:, 0, :].contiguous()
        return out[:, 1:, :].contiguous(), pooled.clone()  # H/14 (model_dist_TVTSv2_ViT_H_14.py:151-153)


class _SimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, eps):
        head = LossHead(a.device)
        an, bn, ai, bi = head.sim(a.contiguous().float(), b.contiguous().float(), eps)
        G, Gb, E = a.shape[0], b.shape[0], a.shape[1]
        x = torch.empty(G, Gb, dtype=torch.float32, device=a.device)
        K.gemm_small(an, bn, x, M=G, N=Gb, K=E, sa=(E, 1), sb=(1, E))
        ctx.save_for_backward(an.clone(), bn.clone(), ai.clone(), bi.clone())
        return x

    @staticmethod
    def backward(ctx, dx):
        an, bn, ai, bi = ctx.saved_tensors
        dx = dx.contiguous().float()
        G, Gb, E = an.shape[0], bn.shape[0], an.shape[1]
        dan, dbn = torch.empty_like(an), torch.empty_like(bn)
        K.gemm_small(dx, bn, dan, M=G, N=E, K=Gb, sa=(Gb, 1), sb=(E, 1))
        K.gemm_small(dx, an, dbn, M=Gb, N=E, K=G, sa=(1, Gb), sb=(E, 1))
        da, db = torch.empty_like(an), torch.empty_like(bn)
        K.l2norm_rows_bwd(dan, an, ai, da)
        K.l2norm_rows_bwd(dbn, bn, bi, db)
        return da, db, None


def sim_matrix(a, b, eps=1e-8):
    """Cosine-similarity matrix with the norm clamp (model_dist_TVTSv2_ViT_B_16.py:119-127), HIP kernels."""
    return _SimFn.apply(a, b, eps)
