"""Step engine of the v1 TVTS model over the same HIP kernels as the v2 engine.

What differs from TVTSv2 (reference paths under v1/):
  video tower   Conv3d TUBELET patch embedding (2 frames x 16 x 16 -> one token) as gather + MFMA GEMM over the kept patches
                only, a tube mask PER TUBE, JOINT space-time attention over all kept tokens of the clip (one FULL attention
                site per block instead of the divided time / space pair), pre-LN blocks with erf-GELU, final norm on every
                token, vid_proj on the CLS token           model/video_encoder.py:78-217, model/model_dist_TVTS.py:143-147
  text tower    Hugging Face DistilBERT: learned word + position embeddings -> LayerNorm, POST-LN blocks (separate q / k / v /
                out projections, erf-GELU FFN, eps 1e-12), padded keys masked, [CLS] row -> ReLU -> Linear
                                                            model/model_dist_TVTS.py:34,65-68,131-141
  sort head     the same SortTransformer, fed the un-projected normed ViT tokens (width 768) and the [CLS] text rows
                                                            model/model_dist_TVTS.py:99-116
Dropout: the reference puts the text tower in training mode (model_dist_TVTS.py:33-34 ``self.text_model.train()``), so every
pretraining step runs DistilBERT's dropouts (transformers modeling_distilbert: ``Embeddings.dropout`` behind the embedding
LayerNorm, ``attention_dropout`` on the softmax probabilities, ``FFN.dropout`` behind lin2; all p = 0.1, config defaults).  Built
here with a COUNTER-BASED generator (splitmix64 of seed + site + element index, tvts_attn_*_len_drop / tvts_dropout_rows): the
attention mask lives inside the attention kernels (probabilities are never materialised), the backward regenerates every mask from
(seed, site), the seed sits in device memory and is advanced by a device op at the start of every training forward (graph replays
draw new masks).  ``engine.training = False`` (the module's eval mode: validation, feature extraction) or
``arch["text_dropout"] = 0`` switch it off.
"""
from __future__ import annotations

import torch

from . import hip as K
from .engine import Engine, ParamStore, _SORT_NAMES

_VIT_NAMES = dict(ln1="norm1", qkv_w="attn.qkv.weight", qkv_b="attn.qkv.bias", o_w="attn.proj.weight", o_b="attn.proj.bias",
                  ln2="norm2", fc_w="mlp.fc1.weight", fc_b="mlp.fc1.bias", pj_w="mlp.fc2.weight", pj_b="mlp.fc2.bias")


class EngineV1(Engine):
    def __init__(self, store: ParamStore):
        super().__init__(store)
        assert self.arch.get("family") == "v1"
        if self.dh_text != 64:
            raise NotImplementedError("the masked FULL attention kernels are used at head dim 64 here")
        self.training = True                                   # the nn.Module wrapper mirrors its own .training flag here
        self.text_drop_p = float(self.arch.get("text_dropout", 0.1))
        # seed of the step's dropout masks (int64 device scalar = the bits of an unsigned 64-bit counter)
        # every data-parallel rank draws its own masks, as the reference's ranks do (each seeds its own generator): the rank is mixed
        # into the seed with an odd 64-bit constant.  The seed is trainer state: Trainer_TVTS saves it with the checkpoint.
        from .dist import world
        seed = (int(self.arch.get("dropout_seed", 0x1234ABCD)) + world()[1] * self.DROP_RANK_STRIDE) & ((1 << 64) - 1)
        self.drop_seed = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64, device=self.dev)
        self._drop_active = 0.0

    DROP_STEP_STRIDE = 0x51ED270B7F4A7C15  # added to the seed (mod 2^64) once per training forward
    DROP_RANK_STRIDE = 0x9E3779B97F4A7C15  # ... and once per rank below this one

    @classmethod
    def _rank_offset(cls, rank):
        return (rank * cls.DROP_RANK_STRIDE) & ((1 << 64) - 1)

    def drop_seed_base(self) -> int:
        """the rank-independent part of the seed (configured seed + step advances, unsigned 64-bit): what a checkpoint stores --
        rank 0 writes the file, every rank reads it"""
        from .dist import world
        return ((int(self.drop_seed.item()) & ((1 << 64) - 1)) - self._rank_offset(world()[1])) & ((1 << 64) - 1)

    def set_drop_seed_base(self, base: int):
        """resume: this rank's seed = stored base + this rank's offset (mod 2^64), so that every rank continues ITS OWN mask
        sequence -- loading rank 0's seed verbatim would give all ranks the same masks"""
        from .dist import world
        seed = ((int(base) & ((1 << 64) - 1)) + self._rank_offset(world()[1])) & ((1 << 64) - 1)
        self.drop_seed.fill_(seed - (1 << 64) if seed >= (1 << 63) else seed)

    def _advance_drop_seed(self):
        """new masks for this step (a device op: a captured graph advances the seed on every replay)"""
        self._drop_active = self.text_drop_p if self.training else 0.0
        if self._drop_active > 0.0:
            self.drop_seed.add_(self.DROP_STEP_STRIDE)

    # ------------------------------------------------------------------ DistilBERT text tower
    def _pln_fwd(self, pre, x, xb, x_out, xb_out, tag, M, N, L, kv_len, layer=0):
        """one POST-LN DistilBERT block: x (fp32) / xb (bf16 copy) -> x_out / xb_out"""
        a, P = self.arch, self.P
        dp = self._drop_active
        Wt, h = a["text_width"], a["text_heads"]
        qkv = self._b(tag + ".qkv", (M, 3 * Wt))
        for i, lin in enumerate(("q_lin", "k_lin", "v_lin")):  # three projections into the packed [M, 3 Wt] buffer
            K.gemm_nt(xb, P.w(pre + f"attention.{lin}.weight"), qkv[:, i * Wt:(i + 1) * Wt], M=M, bias=P.p(pre + f"attention.{lin}.bias"))
        att, lse = self._b(tag + ".att", (M, Wt)), self._f(tag + ".lse", (M, h))
        if dp > 0.0:  # weights = dropout(softmax(scores)) inside the kernel
            K.attn_fwd_len_drop(qkv, kv_len, att, lse, B=N, heads=h, S=L, p=dp, seed=self.drop_seed, site=1 + 2 * layer, head_dim=self.dh_text)
        else:
            K.attn_fwd_len(qkv, kv_len, att, lse, B=N, heads=h, S=L, head_dim=self.dh_text)
        pre1 = self._f(tag + ".pre1", (M, Wt))
        self._lin(att, pre + "attention.out_lin.weight", pre + "attention.out_lin.bias", pre1, M, residual=x)
        x1, x1b = self._f(tag + ".x1", (M, Wt)), self._b(tag + ".x1b", (M, Wt))
        self._ln(pre1, pre + "sa_layer_norm", 1e-12, x1, tag + ".ln_sa")
        K.cast_f32_bf16(x1, x1b)
        hpre, hact = self._b(tag + ".h", (M, a["text_ffn"])), self._b(tag + ".a", (M, a["text_ffn"]))
        self._lin(x1b, pre + "ffn.lin1.weight", pre + "ffn.lin1.bias", hact, M, act="gelu", preact=hpre)
        pre2 = self._f(tag + ".pre2", (M, Wt))
        if dp > 0.0:  # pre2 = dropout(lin2(.)) + x1
            y2 = self._f("txt.s.y2", (M, Wt))
            self._lin(hact, pre + "ffn.lin2.weight", pre + "ffn.lin2.bias", y2, M)
            K.dropout_rows(y2, p=dp, seed=self.drop_seed, site=2 + 2 * layer, residual=x1, out=pre2)
        else:
            self._lin(hact, pre + "ffn.lin2.weight", pre + "ffn.lin2.bias", pre2, M, residual=x1)
        self._ln(pre2, pre + "output_layer_norm", 1e-12, x_out, tag + ".ln_out")
        K.cast_f32_bf16(x_out, xb_out)

    def _pln_bwd(self, pre, xb_in, dx_out, dx_in, tag, M, N, L, kv_len, layer=0):
        """dx_out: fp32 grad wrt the block output; writes the fp32 grad wrt the block input into dx_in."""
        a, P, B_ = self.arch, self.P, self.buf
        dp = self._drop_active
        Wt, h, Ff = a["text_width"], a["text_heads"], a["text_ffn"]
        # output_layer_norm: y = LN(pre2), pre2 = lin2(gelu(lin1(x1))) + x1
        dpre2, dpre2b = self._f("txt.s.dpre2", (M, Wt)), self._b("txt.s.dpre2b", (M, Wt))
        self._ln_bwd(dx_out, B_[tag + ".pre2"], pre + "output_layer_norm", tag + ".ln_out", dpre2, dx_bf16=dpre2b)
        dh = self._b("txt.s.dh", (M, Ff))
        dy2b = dpre2b
        if dp > 0.0:  # gradient of lin2's output: the forward's mask again (the residual path keeps the unmasked dpre2)
            dy2b = self._b("txt.s.dy2b", (M, Wt))
            K.dropout_rows(dpre2, p=dp, seed=self.drop_seed, site=2 + 2 * layer, out_bf16=dy2b)
        self._lin_bwd(dy2b, B_[tag + ".a"], pre + "ffn.lin2.weight", pre + "ffn.lin2.bias", dh, M, gate_h=B_[tag + ".h"], gate_act="gelu")
        dx1 = self._f("txt.s.dx1", (M, Wt))  # = dpre2 (residual) + lin1 dgrad, fused as the GEMM's fp32 residual epilogue
        self._lin_bwd(dh, B_[tag + ".x1b"], pre + "ffn.lin1.weight", pre + "ffn.lin1.bias", dx1, M, residual=dpre2)
        # sa_layer_norm: x1 = LN(pre1), pre1 = out_lin(att) + x
        dpre1, dpre1b = self._f("txt.s.dpre1", (M, Wt)), self._b("txt.s.dpre1b", (M, Wt))
        self._ln_bwd(dx1, B_[tag + ".pre1"], pre + "sa_layer_norm", tag + ".ln_sa", dpre1, dx_bf16=dpre1b)
        datt = self._b("txt.s.datt", (M, Wt))
        self._lin_bwd(dpre1b, B_[tag + ".att"], pre + "attention.out_lin.weight", pre + "attention.out_lin.bias", datt, M)
        dqkv, delta = self._b("txt.s.dqkv", (M, 3 * Wt)), self._f("txt.s.delta", (M, h))
        if dp > 0.0:
            K.attn_bwd_len_drop(B_[tag + ".qkv"], kv_len, datt, B_[tag + ".att"], B_[tag + ".lse"], delta, dqkv, B=N, heads=h, S=L,
                                p=dp, seed=self.drop_seed, site=1 + 2 * layer, head_dim=self.dh_text)
        else:
            K.attn_bwd_len(B_[tag + ".qkv"], kv_len, datt, B_[tag + ".att"], B_[tag + ".lse"], delta, dqkv, B=N, heads=h, S=L,
                           head_dim=self.dh_text)
        # dx = dpre1 (residual) + dq Wq + dk Wk + dv Wv: a chain of fp32-residual epilogues over two ping-pong buffers
        acc, tmp = dpre1, self._f("txt.s.dacc", (M, Wt))
        for i, lin in enumerate(("q_lin", "k_lin", "v_lin")):
            dst = dx_in if i == 2 else (tmp if acc is dpre1 else dpre1)
            self._lin_bwd(dqkv[:, i * Wt:(i + 1) * Wt], xb_in, pre + f"attention.{lin}.weight", pre + f"attention.{lin}.bias", dst, M,
                          residual=acc)
            acc = dst

    def text_forward_v1(self, ids, kv_len, cls_rows, N, L):
        """-> (text_before [N, Wt] fp32 = last hidden state of [CLS], text_emb [N, E] = txt_proj(relu(.)))"""
        a, P = self.arch, self.P
        Wt, M, E = a["text_width"], N * L, a["embed"]
        emb = self._f("txt.emb", (M, Wt))
        K.text_embed(ids, P.p("text_model.embeddings.word_embeddings.weight"),
                     P.p("text_model.embeddings.position_embeddings.weight"), emb, N=N, L=L)
        x, xb = self._f("txt.x0", (M, Wt)), self._b("txt.x0b", (M, Wt))
        self._advance_drop_seed()
        if self._drop_active > 0.0:  # Embeddings: dropout(LayerNorm(word + position))
            xln = self._f("txt.xln", (M, Wt))
            self._ln(emb, "text_model.embeddings.LayerNorm", 1e-12, xln, "txt.ln_emb")
            K.dropout_rows(xln, p=self._drop_active, seed=self.drop_seed, site=0, out=x, out_bf16=xb)
        else:
            self._ln(emb, "text_model.embeddings.LayerNorm", 1e-12, x, "txt.ln_emb")
            K.cast_f32_bf16(x, xb)
        for l in range(a["text_layers"]):
            xo, xbo = self._f(f"txt.x{l + 1}", (M, Wt)), self._b(f"txt.x{l + 1}b", (M, Wt))
            self._pln_fwd(f"text_model.transformer.layer.{l}.", x, xb, xo, xbo, f"txt{l}", M, N, L, kv_len, layer=l)
            x, xb = xo, xbo
        before = self._f("txt.before", (N, Wt))
        K.rows_gather(x, cls_rows, before)
        act = self._f("txt.relu", (N, Wt))
        K.relu(before, act)
        t = self._f("txt.t", (N, E))
        K.gemm_small(act, P.p("txt_proj.1.weight"), t, M=N, N=E, K=Wt, sa=(Wt, 1), sb=(1, Wt), bias=P.p("txt_proj.1.bias"))
        return before, t

    def text_backward_v1(self, dt, ids, kv_len, cls_rows, N, L, tok_sort=None):
        a, P, B_ = self.arch, self.P, self.buf
        Wt, M, E = a["text_width"], N * L, a["embed"]
        K.gemm_small(dt, B_["txt.relu"], P.g("txt_proj.1.weight"), M=E, N=Wt, K=N, sa=(1, E), sb=(Wt, 1), accumulate=True)
        ones = self._ones(N)
        K.gemm_small(ones, dt, P.g("txt_proj.1.bias").view(1, E), M=1, N=E, K=N, sa=(0, 1), sb=(E, 1), accumulate=True)
        dact = self._f("txt.dact", (N, Wt))
        K.gemm_small(dt, P.p("txt_proj.1.weight"), dact, M=N, N=Wt, K=E, sa=(E, 1), sb=(Wt, 1))
        dbefore = self._f("txt.dbefore", (N, Wt))
        K.relu(B_["txt.before"], dbefore, dy=dact)
        dx = self._f("txt.dxA", (M, Wt), zero=True)
        K.rows_gather(dbefore, cls_rows, dx, scatter_add=True)  # only the [CLS] rows of the last hidden state are consumed
        for l in reversed(range(a["text_layers"])):
            nx = "B" if (a["text_layers"] - l) % 2 == 1 else "A"
            dxi = self._f("txt.dx" + nx, (M, Wt))
            self._pln_bwd(f"text_model.transformer.layer.{l}.", B_[f"txt.x{l}b"], dx, dxi, f"txt{l}", M, N, L, kv_len, layer=l)
            dx = dxi
        if self._drop_active > 0.0:  # through the embedding dropout
            ddrop = self._f("txt.s.ddrop", (M, Wt))
            K.dropout_rows(dx, p=self._drop_active, seed=self.drop_seed, site=0, out=ddrop)
            dx = ddrop
        demb = self._f("txt.demb", (M, Wt))
        self._ln_bwd(dx, B_["txt.emb"], "text_model.embeddings.LayerNorm", "txt.ln_emb", demb)
        K.text_embed_bwd(demb, ids, P.g("text_model.embeddings.word_embeddings.weight"),
                         P.g("text_model.embeddings.position_embeddings.weight"), N=N, L=L, tok_sort=tok_sort)

    # ------------------------------------------------------------------ tubelet ViT with joint attention
    def video_forward_v1(self, video, keep, B, tubes, cls_rows):
        """-> (tokens [B*S, W] fp32 after the final norm, video_emb [B, E])"""
        a, P = self.arch, self.P
        W, E, p, tb = a["width"], a["embed"], a["patch"], a["tubelet"]
        n = keep.shape[2]
        S = 1 + tubes * n
        M, Mp = B * S, B * tubes * n
        cols = self._b("vit.im2col", (Mp, P.conv_k))
        K.patch_gather_tube(video, keep, cols, B=B, tubes=tubes, tubelet=tb, n=n, img=a["image"], patch=p)
        pe = self._f("vit.patch", (Mp, W))
        K.gemm_nt(cols, P.w_conv(), pe, M=Mp, bias=P.p("video_model.patch_embed.proj.bias"))
        tok = self._f("vit.x0", (M, W))
        K.vit_assemble(pe, P.p("video_model.cls_token").view(W), P.p("video_model.pos_embed").view(-1, W),
                       P.p("video_model.temporal_embed").view(-1, W), keep, tok, B=B, T=tubes, n=n)
        x = tok
        for l in range(a["layers"]):
            xo = self._f(f"vit.x{l + 1}", (M, W))
            self._block_fwd(f"video_model.blocks.{l}.", _VIT_NAMES, x, xo, f"vit{l}", M, W, a["heads"], B, S, False, "gelu", 1e-6)
            x = xo
        out = self._f("vit.out", (M, W))
        self._ln(x, "video_model.norm", 1e-6, out, "vit.norm")
        cls = self._f("vit.cls", (B, W))
        K.rows_gather(out, cls_rows, cls)
        emb = self._f("mdl.video_emb", (B, E))
        K.gemm_small(cls, P.p("vid_proj.0.weight"), emb, M=B, N=E, K=W, sa=(W, 1), sb=(1, W), bias=P.p("vid_proj.0.bias"))
        return out, emb

    def video_backward_v1(self, dout_b, keep, B, tubes):
        """dout_b: bf16 [B*S, W] grad of the normed tokens (sort head + the CLS rows' share from vid_proj)"""
        a, P, B_ = self.arch, self.P, self.buf
        W = a["width"]
        n = keep.shape[2]
        S = 1 + tubes * n
        M, Mp = B * S, B * tubes * n
        dx, dxb = self._f("vit.dxA", (M, W)), self._b("vit.dxbA", (M, W))
        self._ln_bwd(dout_b, B_[f"vit.x{a['layers']}"], "video_model.norm", "vit.norm", dx, dx_bf16=dxb)
        for l in reversed(range(a["layers"])):
            nx = "B" if (a["layers"] - l) % 2 == 1 else "A"
            dxi, dxbi = self._f("vit.dx" + nx, (M, W)), self._b("vit.dxb" + nx, (M, W))
            self._block_bwd(f"video_model.blocks.{l}.", _VIT_NAMES, B_[f"vit.x{l}"], dx, dxb, dxi, dxbi, f"vit{l}", M, W, a["heads"],
                            B, S, False, "gelu", "vit.s")
            dx, dxb = dxi, dxbi
            self._ready(f"video_model.blocks.{l}.")
        dpatch = self._b("vit.dpatch", (Mp, W))
        K.vit_assemble_bwd(dx, keep, dpatch, P.g("video_model.cls_token").view(W), P.g("video_model.pos_embed").view(-1, W),
                           P.g("video_model.temporal_embed").view(-1, W), B=B, T=tubes, n=n)
        K.gemm_tn(dpatch, B_["vit.im2col"], P.g2d("video_model.patch_embed.proj.weight"), M=Mp, accumulate=True,
                  colsum=P.g("video_model.patch_embed.proj.bias"))
        self._ready("video_model.cls_token", "video_model.pos_embed", "video_model.temporal_embed", "video_model.patch_embed.")
        self._ready("video_model.norm.")

    # ------------------------------------------------------------------ whole model
    def prepare_batch(self, data: dict):
        """v1 batch dict (v1/trainer/trainer.py:121-131): text = the tokenizer's {'input_ids', 'attention_mask'} (right-padded),
        video fp32 [B, T, 3, H, W], keep_ind int64 [B, n_tubes, n_keep] (one mask per tube)."""
        a = self.arch
        video = self._clip_to_device(data["video"]).to(torch.float32).contiguous()
        B, T = video.shape[:2]
        tubes = T // a["tubelet"]
        ids = data["text"]["input_ids"].detach().to("cpu", torch.int64)
        mask = data["text"]["attention_mask"].detach().to("cpu", torch.int64)
        lens = mask.sum(-1)
        assert bool((mask == (torch.arange(mask.shape[1])[None] < lens[:, None])).all()), "attention_mask must be a right-padded prefix mask"
        N, L = ids.shape
        L = int(lens.max())
        NT = N // B
        keep = data["keep_ind"][:, :tubes].to(torch.int32).contiguous().to(self.dev)
        n = keep.shape[2]
        S = 1 + tubes * n
        So = S + NT
        return dict(video=video, ids=ids[:, :L].to(torch.int32).contiguous().to(self.dev), kv_len=lens.to(torch.int32).to(self.dev),
                    tok_sort=tuple(t.to(self.dev) for t in K.token_sort(ids[:, :L])),  # ordered word-embedding gradient sums
                    txt_cls_rows=(torch.arange(N) * L).to(torch.int32).to(self.dev), keep=keep, B=B, T=T, tubes=tubes, N=N, NT=NT, L=L,
                    n=n, S=S, vid_rows=(torch.arange(B) * S).to(torch.int32).to(self.dev),
                    sort_rows=(torch.arange(B)[:, None] * So + S + torch.arange(NT)[None, :]).reshape(-1).to(torch.int32).to(self.dev),
                    sort_rows64=(torch.arange(B)[:, None] * So + S + torch.arange(NT)[None, :]).reshape(-1).to(self.dev))

    def forward(self, pb: dict):
        a = self.arch
        self.ctx = pb
        self._tick += 1  # (workspace: a buffer whose shape changes from here on belongs to a new step, see _b)
        B, N, NT, L, S, E = pb["B"], pb["N"], pb["NT"], pb["L"], pb["S"], a["embed"]
        before, t = self.text_forward_v1(pb["ids"], pb["kv_len"], pb["txt_cls_rows"], N, L)
        text_emb = self._f("mdl.text_emb", (B, E))
        tmean_scratch = self._f("mdl.text_before_e", (B, NT, E))
        K.text_mean(t, text_emb, tmean_scratch, NT=NT, B=B)          # mean over the NT captions (model_dist_TVTS.py:104-107)
        text_before = self._f("mdl.text_before", (B, NT, a["text_width"]))
        K.text_mean(before, self._f("mdl.tb_mean", (B, a["text_width"])), text_before, NT=NT, B=B)  # [NT,B,W] -> [B,NT,W] (:99-101)
        out, video_emb = self.video_forward_v1(pb["video"], pb["keep"], B, pb["tubes"], pb["vid_rows"])
        if self.embeds_ready is not None:
            self.embeds_ready(text_emb, video_emb)
        pred = self.sort_forward(out, text_before, B, S, NT) if NT != 1 else None
        return text_emb, video_emb, pred

    def backward(self, d_text, d_video, d_pred):
        a, pb, P = self.arch, self.ctx, self.P
        B, N, NT, L, S, E, W = pb["B"], pb["N"], pb["NT"], pb["L"], pb["S"], a["embed"], a["width"]
        if d_text is not None:
            dt = self._f("mdl.dt", (N, E))
            K.text_mean_bwd(d_text, dt, NT=NT, B=B)
            self.text_backward_v1(dt, pb["ids"], pb["kv_len"], pb["txt_cls_rows"], N, L, tok_sort=pb.get("tok_sort"))
            self._ready("text_model.")
            self._ready("txt_proj.")
        # vid_proj on the CLS token: dW += d_video^T cls, db += colsum, d_cls = d_video W
        cls = self.buf["vit.cls"]
        K.gemm_small(d_video, cls, P.g("vid_proj.0.weight"), M=E, N=W, K=B, sa=(1, E), sb=(W, 1), accumulate=True)
        ones = self._ones(B)
        K.gemm_small(ones, d_video, P.g("vid_proj.0.bias").view(1, E), M=1, N=E, K=B, sa=(0, 1), sb=(E, 1), accumulate=True)
        dcls = self._f("vit.dcls", (B, W))
        K.gemm_small(d_video, P.p("vid_proj.0.weight"), dcls, M=B, N=W, K=E, sa=(E, 1), sb=(W, 1))
        self._ready("vid_proj.")
        dout = self._b("mdl.dout", (B * S, W))
        if d_pred is not None:
            dxs = self.sort_backward(d_pred, B, S, NT)
            K.sort_assemble_bwd(dxs, dcls, dout, P.g("pred_model.type_embed").view(2, W), B=B, S=S, off=0, Sv=S, NT=NT)
            self._ready("pred_model.")
        else:
            K.sort_assemble_bwd(None, dcls, dout, None, B=B, S=S, off=0, Sv=S, NT=NT)
        self.video_backward_v1(dout, pb["keep"], B, pb["tubes"])
