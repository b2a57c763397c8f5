"""Which rows of a bf16 residual stream carry the loss error?  (round-5 question)
fp32 CPU oracle with the space-time blocks' residual stream rounded to bf16 where the engine's --bf16-residual rounds it
(s_res and the block output; t_res is bf16 in every mode) -- on every row, on every row but the CLS row, on the CLS row only.
Prints |d loss1| against the unrounded oracle on the configuration that fails the 1e-2 gate (H/14-style toy, 3 pairs, 16 frames,
NT = 1) and on a B-style toy, over several parameter seeds."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tvts_oracle as O
from tvts_amd import arch as A

MODE = ["none"]
_orig = O.st_block


def rnd(x):
    x = RoundGrad.apply(x)
    if MODE[0] in ("none", "tres"):
        return x
    y = x.to(torch.bfloat16).float()
    if MODE[0] == "all":
        return y
    keep_cls = MODE[0] in ("patches", "patches_lnin")  # round patch rows only
    out = y.clone() if keep_cls else x.clone()
    out[:, 0] = x[:, 0] if keep_cls else y[:, 0]
    return out


GMODE = ["none"]


class RoundGrad(torch.autograd.Function):
    """identity forward; the gradient of the residual stream is rounded to bf16 on its way back (all rows / patch rows only)"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if GMODE[0] == "none":
            return g
        y = g.to(torch.bfloat16).float()
        if GMODE[0] == "patches":
            y = y.clone(); y[:, 0] = g[:, 0]
        return y


def rnd_ln_in(x):
    """LayerNorm INPUT of the CLS row rounded (the stream itself stays exact): mode cls_ln_bf16"""
    if MODE[0] != "patches_lnin":
        return x
    return x.to(torch.bfloat16).float()


def st_block(x, P, pre, arch, T, n):
    h = arch["heads"]
    lin = O.linear
    t_out = O.divided_attention(O.layer_norm(rnd_ln_in(x), P[pre + "ln_3.weight"], P[pre + "ln_3.bias"], 1e-5), P[pre + "timeattn.qkv.weight"],
                                P[pre + "timeattn.qkv.bias"], P[pre + "timeattn.proj.weight"], P[pre + "timeattn.proj.bias"], h, "time", T, n, lin)
    t_res = (x + t_out).to(torch.bfloat16).float() if MODE[0] != "none" else x + t_out  # bf16 in every engine mode
    s_out = O.divided_attention(O.layer_norm(t_res, P[pre + "ln_1.weight"], P[pre + "ln_1.bias"], 1e-5), P[pre + "attn.qkv.weight"],
                                P[pre + "attn.qkv.bias"], P[pre + "attn.proj.weight"], P[pre + "attn.proj.bias"], h, "space", T, n, lin)
    s_res = rnd(x + s_out)
    hid = O._act(arch)(lin(O.layer_norm(rnd_ln_in(s_res), P[pre + "ln_2.weight"], P[pre + "ln_2.bias"], 1e-5), P[pre + "mlp.c_fc.weight"], P[pre + "mlp.c_fc.bias"]))
    return rnd(s_res + lin(hid, P[pre + "mlp.c_proj.weight"], P[pre + "mlp.c_proj.bias"]))


O.st_block = st_block
torch.set_num_threads(8)
CFG = (("H/14-style toy, 3 pairs, T=16, NT=1 (the failing gate)", A.small_arch_h(num_frames=16), dict(B=3, T=16, n_trans=1, caption_len=9)),
       ("H/14-style toy, 12 layers", A.small_arch_h(layers=12, num_frames=16), dict(B=3, T=16, n_trans=1, caption_len=9)),
       ("B-style toy, 12 layers, 4 pairs, T=3, NT=4", A.small_arch(layers=12), dict(B=4, T=3, caption_len=11)))
for label, a, kw in CFG:
    oarch = O.tiny_arch(**a)
    res = {m: [] for m in ("tres", "all", "patches", "patches_lnin", "cls")}
    gres = {m: [] for m in ("all", "patches")}
    for seed in range(6):
        P = O.synth_params(oarch, seed=seed)
        b = O.synth_batch(oarch, seed=100 + seed, **kw)
        with torch.no_grad():
            MODE[0] = "none"; O.st_block = _orig
            l0 = float(O.step_losses(P, b, oarch)[0])
            O.st_block = st_block
            for m in res:
                MODE[0] = m
                res[m].append(abs(float(O.step_losses(P, b, oarch)[0]) - l0))
        # gradient stream: forward exact (MODE none), gradient rounded
        MODE[0] = "none"
        def grads(gm):
            GMODE[0] = gm
            L = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            l1, l2, *_ = O.step_losses(L, b, oarch)
            (l1 + l2).backward()
            return {k: v.grad for k, v in L.items() if v.grad is not None}
        g0 = grads("none")
        for gm in gres:
            g = grads(gm)
            row = {}
            for k in ("video_model.positional_embedding", "video_model.class_embedding", "video_model.conv1.weight",
                      "video_model.transformer.resblocks.0.mlp.c_fc.weight"):
                row[k] = float((g[k] - g0[k]).norm() / g0[k].norm())
            tot = sum(float((g[k] - g0[k]).norm()) ** 2 for k in g0) ** 0.5 / sum(float(g0[k].norm()) ** 2 for k in g0) ** 0.5
            row["all tensors"] = tot
            gres[gm].append(row)
        GMODE[0] = "none"
    print(label)
    for m, v in res.items():
        t = torch.tensor(v)
        print(f"   forward stream, bf16 rounding of {m:13s}: |d loss1| mean {t.mean():.2e} max {t.max():.2e}")
    for gm, rows in gres.items():
        keys = rows[0].keys()
        print(f"   gradient stream, bf16 rounding of {gm:8s}: rel-L2 error of " + ", ".join(f"{k.split('.')[-2] if '.' in k else k}.{k.split('.')[-1]} {max(r[k] for r in rows):.2e}" for k in keys))
