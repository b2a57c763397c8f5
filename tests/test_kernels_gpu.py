"""GPU parity of every C-ABI kernel against plain fp32 torch-CPU / oracle restatements (run with -m gpu)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tvts_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tvts_amd import hip
    return hip


DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ probes
def test_tr16_read_semantics(K):
    """ds_read_b64_tr_b16: lane (l&15) of each 16-lane group receives column (l&15) of that group's 4x16 block."""
    x = torch.arange(16 * 64, dtype=torch.float32).reshape(16, 64) % 251
    out = torch.zeros(64 * 4, dtype=torch.bfloat16, device=DEV)
    K.probe_tr16(bf(x).to(DEV), out)
    got = out.float().cpu().reshape(64, 4)
    exp = torch.empty(64, 4)
    for l in range(64):
        for e in range(4):
            exp[l, e] = bf(x)[(l >> 4) * 4 + e, l & 15].float()
    assert torch.equal(got, exp), (got[:20], exp[:20])


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (300, 256, 128), (1000, 768, 768), (77, 2304, 256), (9420, 768, 3072)])
def test_gemm_nt_plain(K, M, N, K_):
    a, b = bf(rnd(M, K_, seed=1)), bf(rnd(N, K_, seed=2) * K_ ** -0.5)
    ref = a.float() @ b.float().t()
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 4e-3)):
        out = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
        K.gemm_nt(a.to(DEV), b.to(DEV), out)
        assert rel(out.float(), ref) < tol, (dt, rel(out.float(), ref))


def test_gemm_nt_transpose_detecting(K):
    """A = [I | 0] with an asymmetric B: catches swapped rows/cols in the C write."""
    M = N = 128
    K_ = 128
    a = torch.zeros(M, K_); a[:, :M] = torch.eye(M)
    b = rnd(N, K_, seed=3)
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    K.gemm_nt(bf(a).to(DEV), bf(b).to(DEV), out)
    assert rel(out, bf(b).float()[:, :M].t()) < 1e-6


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_gemm_nt_epilogues(K, act):
    M, N, K_ = 333, 512, 256
    a, b = bf(rnd(M, K_, seed=4)), bf(rnd(N, K_, seed=5) * K_ ** -0.5)
    bias, res = rnd(N, seed=6), rnd(M, N, seed=7)
    fn = O.quick_gelu if act == "quick_gelu" else O.gelu_erf
    pre = a.float() @ b.float().t() + bias
    # bias + activation + pre-activation side output
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    h = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt(a.to(DEV), b.to(DEV), out, bias=bias.to(DEV), act=act, preact=h)
    assert rel(h.float(), pre) < 4e-3 and rel(out.float(), fn(pre)) < 5e-3
    # bias + fp32 residual, fp32 out
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    K.gemm_nt(a.to(DEV), b.to(DEV), out32, bias=bias.to(DEV), residual=res.to(DEV))
    assert rel(out32, pre + res) < 2e-5
    # activation-gradient gate: out = (a b^T) * act'(h)
    hh = bf(rnd(M, N, seed=8))
    x = hh.float().clone().requires_grad_(True)
    fn(x).sum().backward()
    g = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt(a.to(DEV), b.to(DEV), g, gate_h=hh.to(DEV), gate_act=act)
    assert rel(g.float(), (a.float() @ b.float().t()) * x.grad) < 5e-3
    # TVTS_GEMM_SIDE_DERIV (round 5): the forward stores act'(x) in place of the pre-activation, the gate multiplies by it as is --
    # on every tile kernel (256 x 256 with the generic and the hand-scheduled epilogue, 128 x 128, the ring forms)
    xp = pre.clone().requires_grad_(True)
    fn(xp).sum().backward()
    for M2, tile in ((M, None), (M, 128), (1024, 256), (1024, None)):
        a2 = bf(rnd(M2, K_, seed=9)).to(DEV)
        pre2 = a2.float().cpu() @ b.float().t() + bias
        xp2 = pre2.clone().requires_grad_(True)
        fn(xp2).sum().backward()
        o2, d2 = (torch.empty(M2, N, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        K.gemm_nt(a2, b.to(DEV), o2, bias=bias.to(DEV), act=act, preact=d2, side_deriv=True, tile=tile)
        assert rel(o2.float(), fn(pre2)) < 5e-3 and rel(d2.float(), xp2.grad) < 5e-3, (M2, tile)
        g2 = torch.empty(M2, N, dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(a2, b.to(DEV), g2, gate_h=d2, gate_act=act, side_deriv=True, tile=tile)
        assert rel(g2.float(), (a2.float().cpu() @ b.float().t()) * d2.float().cpu()) < 5e-3, (M2, tile)


@pytest.mark.parametrize("M,Na,Nb", [(64, 128, 128), (200, 256, 128), (1000, 768, 256), (9420, 768, 768), (37, 512, 128)])
def test_gemm_tn(K, M, Na, Nb):
    p, q = bf(rnd(M, Na, seed=9)), bf(rnd(M, Nb, seed=10))
    ref = p.float().t() @ q.float()
    out = torch.full((Na, Nb), 7.0, dtype=torch.float32, device=DEV)
    K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=False)
    assert rel(out, ref) < 3e-5, rel(out, ref)
    cs = torch.ones(Na, device=DEV)
    K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=True, colsum=cs)
    assert rel(out, 2 * ref) < 3e-5
    assert rel(cs, 1 + p.float().sum(0)) < 1e-5, rel(cs, 1 + p.float().sum(0))
    cs2 = torch.ones(Na, device=DEV)  # the bias gradient is an ordered sum of per-range partials: bit-reproducible
    K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=True, colsum=cs2)
    assert torch.equal(cs, cs2)
    cs3 = torch.ones(Na, device=DEV)
    K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=False, colsum=cs3, workspace=False)  # fallback: atomics (or the one range's owner)
    assert rel(cs3, 1 + p.float().sum(0)) < 1e-5
    out.copy_(2 * ref)
    K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=True, workspace=False)  # fp32-atomic fallback (no workspace)
    assert rel(out, 3 * ref) < 3e-5
    K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=False, workspace=False)
    assert rel(out, ref) < 3e-5


@pytest.mark.parametrize("M,Na,Nb", [(64, 256, 256), (200, 512, 256), (1000, 768, 256), (9420, 768, 768), (37, 512, 128), (4097, 1280, 640),
                                     (40001, 2304, 768), (333, 248, 264)])
def test_gemm_tn_256_tile(K, M, Na, Nb):
    """The pipelined 256x256 weight-gradient kernel (forced; the dispatcher picks it for M >= 32768 and >= 1.5 M outputs): whole
    and ragged tiles, m-ranges that end inside a stage, the fused bias gradient, workspace and fp32-atomic partials, and bit
    equality of the workspace path with itself across two launches (deterministic reduce)."""
    p, q = bf(rnd(M, Na, seed=19)), bf(rnd(M, Nb, seed=20))
    ref = p.float().t().double() @ q.float().double()
    with K.options(tn_tile=256):
        assert K.gemm_tn_select(M, Na, Nb) == 256
        out = torch.full((Na, Nb), float("nan"), dtype=torch.float32, device=DEV)
        K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=False)
        assert rel(out, ref) < 3e-5, rel(out, ref)
        out2 = torch.full((Na, Nb), float("nan"), dtype=torch.float32, device=DEV)
        K.gemm_tn(p.to(DEV), q.to(DEV), out2, accumulate=False)
        assert torch.equal(out, out2)
        cs = torch.ones(Na, device=DEV)
        K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=True, colsum=cs)
        assert rel(out, 2 * ref) < 3e-5
        assert rel(cs, 1 + p.float().sum(0)) < 2e-5, rel(cs, 1 + p.float().sum(0))
        cs2 = torch.ones(Na, device=DEV)
        K.gemm_tn(p.to(DEV), q.to(DEV), out2, accumulate=False, colsum=cs2)
        assert torch.equal(cs, cs2)  # ordered per-range partials of the bias gradient
        cs3 = torch.ones(Na, device=DEV)
        K.gemm_tn(p.to(DEV), q.to(DEV), out2, accumulate=False, colsum=cs3, workspace=False)
        assert rel(cs3, 1 + p.float().sum(0)) < 2e-5
        K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=True, workspace=False)  # fp32-atomic fallback (no workspace)
        assert rel(out, 3 * ref) < 3e-5
        K.gemm_tn(p.to(DEV), q.to(DEV), out, accumulate=False, workspace=False)
        assert rel(out, ref) < 3e-5


@pytest.mark.parametrize("M,N,K_", [(9420, 768, 768), (9420, 2304, 768), (9420, 768, 3072), (18840, 3072, 768), (5000, 1280, 1280), (9432, 768, 2304)])
def test_gemm_nt_streamk(K, M, N, K_):
    """The stream-K walk of the 256x256 NT kernel (TVTS_GEMM_STREAMK; round 4, for the reference's per-GPU batches: M = 12 / 24 x 785):
    work split by K stage inside each XCD's tile range, pieces that do not cover a tile's K leave as fp32 partials, the block that
    arrives last adds them in k order and runs the fused epilogue.  Every epilogue form against an fp64 product, and the same bits
    launch after launch (with other launches in between: the arrival counters are zero again after every call)."""
    a, b = bf(rnd(M, K_, seed=71)).to(DEV), bf(rnd(N, K_, seed=72) * K_ ** -0.5).to(DEV)
    bias = rnd(N, seed=73).to(DEV)
    ref = a.double() @ b.double().t() + bias.double()
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    res = rnd(M, N, seed=74).to(DEV)
    K.gemm_nt(a, b, out, bias=bias, residual=res, streamk=True)
    assert rel(out, ref + res.double()) < 2e-5, rel(out, ref + res.double())
    outb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt(a, b, outb, bias=bias, streamk=True)
    other = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt(a, b, other, streamk=True)
    outb2 = torch.empty_like(outb)
    K.gemm_nt(a, b, outb2, bias=bias, streamk=True)
    assert torch.equal(outb, outb2) and rel(outb.float(), ref) < 4e-3
    # activation + pre-activation side output, and the activation-gradient gate
    pre, act = torch.empty_like(outb), torch.empty_like(outb)
    K.gemm_nt(a, b, act, bias=bias, act="quick_gelu", preact=pre, streamk=True)
    z = ref
    assert rel(pre.float(), z) < 4e-3 and rel(act.float(), z * torch.sigmoid(1.702 * z)) < 5e-3
    gate = torch.empty_like(outb)
    K.gemm_nt(a, b, gate, gate_h=pre, gate_act="quick_gelu", streamk=True)
    h = pre.float().double()
    sg = torch.sigmoid(1.702 * h)
    assert rel(gate.float(), (ref - bias.double()) * (sg * (1 + 1.702 * h * (1 - sg)))) < 6e-3
    ws = K._nt_workspace(a.device)
    assert int(ws[:65536].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("ring", ["ring2", "ring3", "ring4", "ring"])
@pytest.mark.parametrize("M,N,K_", [(5856, 768, 768), (5856, 768, 3072), (1536, 512, 2048), (976, 2304, 768), (1537, 520, 64), (100, 132, 128),
                                    (12001, 768, 192), (3072, 2048, 512), (64, 4, 64)])
def test_gemm_nt_ring_gives_the_bits_of_the_128_kernel(K, M, N, K_, ring):
    """The small-batch form of the 128-column NT kernel (TVTS_GEMM_RING; round 4): one block of 8 waves per CU, a ring of 64-deep
    stages with three of them in flight, epilogue inputs requested ahead by hand-counted loads, 128 / 192 / 256 tile rows.  Same
    tile walk, same k order and the same epilogue arithmetic as the double-buffered 128 kernel: every epilogue form gives the same
    bits, on the step's shapes at 2 / 12 / 24 pairs per GPU, on ragged shapes (rows / columns sticking out of the last tile, K of
    one and two stages -- shorter than the ring) and launch after launch."""
    a, b = bf(rnd(M, K_, seed=81)).to(DEV), bf(rnd(N, K_, seed=82) * K_ ** -0.5).to(DEV)
    bias = rnd(N, seed=83).to(DEV)
    res = rnd(M, N, seed=84).to(DEV)
    gh = bf(rnd(M, N, seed=85)).to(DEV)
    def run(tile):
        outs = []
        o = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        K.gemm_nt(a, b, o, bias=bias, residual=res, tile=tile); outs.append(o)                      # fp32 + residual
        o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(a, b, o, tile=tile); outs.append(o)                                                # plain bf16, no bias
        o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(a, b, o, bias=bias, residual=res, tile=tile); outs.append(o)                      # bf16 + fp32 residual
        o, pre = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(a, b, o, bias=bias, act="quick_gelu", preact=pre, tile=tile); outs += [o, pre]     # activation + pre-activation
        o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(a, b, o, bias=bias, act="gelu", tile=tile); outs.append(o)
        for ga in ("quick_gelu", "gelu", "add"):                                                      # activation-gradient gates, bf16 residual
            o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            K.gemm_nt(a, b, o, gate_h=gh, gate_act=ga, tile=tile); outs.append(o)
        torch.cuda.synchronize()
        return outs
    ref = run("128noring")
    got = run(ring)
    again = run(ring)
    for i, (x, y, z) in enumerate(zip(ref, got, again)):
        assert not torch.isnan(x.float()).any()
        assert torch.equal(y, z)
        assert torch.equal(x, y), i  # (the erf-GELU forms too: their fp contraction is pinned by hand in common.h)
    exact = a.double() @ b.double().t() + bias.double() + res.double()
    assert rel(got[0], exact) < 2e-5


def test_gemm_nt_ring_is_what_the_small_batches_take(K):
    """The dispatcher's choice (tvts_gemm_nt_select): the N = 768 GEMMs of the video tower at the reference's 12 pairs per GPU and the
    text tower up to 24 pairs run on the ring kernel, the wide outputs and every shape of the 192-pair step keep the 256 kernel, a
    forced tile size or TVTS_GEMM_NO_RING means the old kernels."""
    assert K.gemm_nt_select(5856, 768) == 1192 and K.gemm_nt_select(1536, 512) == 1128 and K.gemm_nt_select(3072, 2048) == 1192
    assert K.gemm_nt_select(5856, 2304) == 256 and K.gemm_nt_select(11712, 768) == 256
    assert K.gemm_nt_select(9420, 768) == 128  # the bench's 12 pairs x 785 rows: 444 tiles fill the 128 kernel's 512 slots
    assert K.gemm_nt_select(93696, 768) == 256 and K.gemm_nt_select(24576, 512) == 256
    assert K.gemm_nt_select(5856, 768, tile=128) == 128 and K.gemm_nt_select(5856, 768, tile="noring") == 128
    assert K.gemm_nt_select(93696, 768, tile="ring") == 1128 + 64 * (K.gemm_nt_select(93696, 768, tile="ring") == 1192)


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M,Na,Nb", [(9420, 768, 768), (9420, 2304, 768), (18840, 768, 3072), (4097, 1280, 640), (40001, 768, 768), (3000, 248, 264)])
def test_gemm_tn_fused_reduce_gives_the_bits_of_the_reduce_pass(K, M, Na, Nb, tile):
    """The m-range partials of a weight gradient meet either in tn_reduce_kernel or -- fused, round 4 -- in the block that arrives
    last at the output tile (agent-scope partial stores, an arrival counter per tile).  Both add the ranges in range order on top
    of the output, so the results are the same bits: output (accumulating and not), bias gradient, launch after launch, at the
    reference's per-GPU batches (M = 12 / 24 x 785) and on ragged shapes."""
    p, q = bf(rnd(M, Na, seed=29)).to(DEV), bf(rnd(M, Nb, seed=30)).to(DEV)
    base = rnd(Na, Nb, seed=31).to(DEV)
    res = {}
    for fused in (False, True):
        with K.options(tn_tile=tile, tn_splits=5):
            out = base.clone()
            cs = torch.ones(Na, device=DEV)
            K.gemm_tn(p, q, out, accumulate=True, colsum=cs, fused=fused)
            out2 = torch.full((Na, Nb), float("nan"), dtype=torch.float32, device=DEV)
            K.gemm_tn(p, q, out2, accumulate=False, fused=fused)
            for _ in range(3):  # the counters are zero again after every launch
                out3 = torch.full((Na, Nb), float("nan"), dtype=torch.float32, device=DEV)
                K.gemm_tn(p, q, out3, accumulate=False, fused=fused)
                assert torch.equal(out2, out3)
        torch.cuda.synchronize()
        res[fused] = (out, cs, out2)
    ref = p.float().t().double() @ q.float().double()
    assert rel(res[True][2], ref) < 3e-5
    for x, y in zip(res[False], res[True]):
        assert torch.equal(x, y)
    assert int(K._tn_counters(K._tn_workspace(p.device)).abs().sum()) == 0


def test_gemm_tn_grouped_gives_the_bits_of_the_single_launches(K):
    """tvts_gemm_tn_bf16_grouped: the six weight gradients of a ViT block (M = 12 x 785 token rows, the reference's per-GPU batch) in
    one launch + one reduce launch.  With the same range count every problem's output and bias gradient are the bits of its own
    tvts_gemm_tn_bf16 launch; the automatic plan holds the fp32 tolerance; the plan is re-used by later runs."""
    M, W = 9420, 768
    shapes = [(W, 4 * W), (4 * W, W), (W, W), (3 * W, W), (W, W), (3 * W, W)]
    ps = [bf(rnd(M, na, seed=90 + i)).to(DEV) for i, (na, nb) in enumerate(shapes)]
    qs = [bf(rnd(M, nb, seed=190 + i)).to(DEV) for i, (na, nb) in enumerate(shapes)]
    base = [rnd(na, nb, seed=290 + i).to(DEV) for i, (na, nb) in enumerate(shapes)]

    def problems(outs, css):
        return [dict(p=p, q=q, out=o, M=M, accumulate=True, colsum=c) for p, q, o, c in zip(ps, qs, outs, css)]
    ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=DEV)
    outs = [b.clone() for b in base]
    css = [torch.ones(na, device=DEV) if i % 2 == 0 else None for i, (na, nb) in enumerate(shapes)]
    grp = K.TnGroup(problems(outs, css), ws, splits=4)
    grp.run()
    torch.cuda.synchronize()
    for i, (na, nb) in enumerate(shapes):
        o = base[i].clone()
        c = torch.ones(na, device=DEV) if css[i] is not None else None
        K.gemm_tn(ps[i], qs[i], o, M=M, accumulate=True, colsum=c, tile=128, splits=4, fused=False)
        assert torch.equal(o, outs[i]), i
        if c is not None:
            assert torch.equal(c, css[i]), i
    for o, b in zip(outs, base):
        o.copy_(b)
    grp.run()   # the uploaded plan again
    for i in range(len(shapes)):
        o = base[i].clone()
        K.gemm_tn(ps[i], qs[i], o, M=M, accumulate=True, tile=128, splits=4, fused=False)
        assert torch.equal(o, outs[i])
    outs2 = [b.clone() for b in base]
    K.TnGroup(problems(outs2, [None] * 6), ws).run()   # automatic range count
    for i in range(len(shapes)):
        ref = base[i].cpu().double() + ps[i].cpu().float().t().double() @ qs[i].cpu().float().double()
        assert rel(outs2[i], ref) < 3e-5, (i, rel(outs2[i], ref))
    # the FIRST run of a plan on a non-default, non-blocking stream (the side-stream warm-up in front of a graph capture): the plan's
    # upload is a copy ON THAT STREAM (round 4: a synchronous copy on the legacy stream, ordered only against torch's default stream
    # -- an empty plan on another stream skips every weight gradient silently); and on that stream inside a capture
    side = torch.cuda.Stream()
    outs3 = [b.clone() for b in base]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        busy = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
        for _ in range(8):
            busy.zero_()  # work queued in front of the upload on the same stream
        g3 = K.TnGroup(problems(outs3, [None] * 6), ws, splits=4)
        g3.run()
    side.synchronize()
    for i in range(len(shapes)):
        assert torch.equal(outs3[i], outs[i]), i
    outs4 = [b.clone() for b in base]
    g4 = K.TnGroup(problems(outs4, [None] * 6), ws, splits=4)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g4.run()  # first run = upload, captured as a copy node from the plan's page-locked staging memory
    graph.replay()
    torch.cuda.synchronize()
    for i in range(len(shapes)):
        assert torch.equal(outs4[i], outs[i]), i


@pytest.mark.parametrize("R,N,Kd,S", [(12, 768, 768, 785), (192, 768, 3072, 785), (2, 1280, 5120, 1217), (5, 64, 96, 3), (48, 1280, 1280, 7)])
def test_rows_linear(K, R, N, Kd, S):
    """tvts_rows_linear_bf16: a few STRIDED rows (the CLS token of every clip: row stride S * K of the block's [B * S, K] operand)
    through a linear layer with fp32 result and fp32 residual -- the hybrid residual stream's fix-up of `x + proj(...)` -- against the
    fp64 product of the same bf16 operands."""
    a_all = bf(rnd(R * S, Kd, seed=70)).to(DEV)
    w = bf(rnd(N, Kd, seed=71) * 0.05).to(DEV)
    bias, res = rnd(N, seed=72).to(DEV), rnd(R, N, seed=73).to(DEV)
    a = a_all.view(R, S * Kd)[:, :Kd]
    out = torch.full((R, N), float("nan"), device=DEV)
    K.rows_linear(a, w, out, bias=bias, residual=res)
    ref = res.double().cpu() + bias.double().cpu() + a.float().cpu().double() @ w.float().cpu().double().t()
    assert rel(out, ref) < 2e-6, rel(out, ref)
    out2 = torch.full((R, N), float("nan"), device=DEV)
    K.rows_linear(a, w, out2)
    assert rel(out2, a.float().cpu().double() @ w.float().cpu().double().t()) < 2e-6
    K.rows_linear(a, w, out2, bias=bias, residual=res)
    assert torch.equal(out, out2)  # fixed summation order


def test_gemm_tn_tile_selection(K):
    """The plan of the weight-gradient entry point (a cost model of tile and range count, csrc/gemm.hip; tools/tn_plan_check.py
    measures it against both tiles' best): the long contractions of the 192-pair step take the 256x256 kernel -- the text tower's
    included --, short contractions and small outputs the 128x128 one; at the reference's 24 pairs per GPU the wide ViT gradients
    already take the 256x256 kernel, at 12 pairs none does."""
    M = 192 * 785
    assert K.gemm_tn_select(M, 2304, 768) == 256 and K.gemm_tn_select(M, 768, 3072) == 256
    assert K.gemm_tn_select(M, 768, 768) == 256 and K.gemm_tn_select(24576, 2048, 512) == 256
    assert K.gemm_tn_select(24576, 512, 512) == 128 and K.gemm_tn_select(256, 2048, 512) == 128
    assert K.gemm_tn_select(24 * 785, 2304, 768) == 256 and K.gemm_tn_select(24 * 785, 768, 768) == 128
    assert K.gemm_tn_select(12 * 785, 2304, 768) == 128
    assert K.gemm_tn_select(12 * 785, 2304, 768, tile=256) == 256 and K.gemm_tn_select(M, 2304, 768, tile=128) == 128


def test_gemm_tn_views(K):
    """P and Q as column slices of wider buffers (leading dimension != width)."""
    M = 500
    big = bf(rnd(M, 768, seed=11))
    p, q = big[:, 128:384], big[:, 512:640]
    out = torch.zeros(256, 128, dtype=torch.float32, device=DEV)
    bd = big.to(DEV)
    K.gemm_tn(bd[:, 128:384], bd[:, 512:640], out, accumulate=False)
    assert rel(out, p.float().t() @ q.float()) < 3e-5


def test_gemm_small_and_colsum(K):
    a, b = rnd(37, 50, seed=12), rnd(50, 23, seed=13)
    out = torch.zeros(37, 23, device=DEV)
    K.gemm_small(a.to(DEV), b.to(DEV), out, M=37, N=23, K=50, sa=(50, 1), sb=(23, 1), alpha=0.5)
    assert rel(out, 0.5 * a @ b) < 1e-5
    out2 = torch.ones(50, 23, device=DEV)  # A^T via strides, accumulate
    c = rnd(37, 23, seed=14)
    K.gemm_small(a.to(DEV), c.to(DEV), out2, M=50, N=23, K=37, sa=(1, 50), sb=(23, 1), accumulate=True)
    assert rel(out2, 1 + a.t() @ c) < 1e-5
    x = bf(rnd(1001, 512, seed=15))
    s = torch.ones(512, device=DEV)
    K.colsum(x.to(DEV), s)
    assert rel(s, 1 + x.float().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("W,eps", [(768, 1e-5), (512, 1e-6), (128, 1e-5), (1280, 1e-5)])
def test_layernorm(K, W, eps):
    M = 301
    x = rnd(M, W, seed=16) * 2 + 0.3
    g, b = 1 + 0.1 * rnd(W, seed=17), 0.1 * rnd(W, seed=18)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = O.layer_norm(xr, gr, br, eps)
    dy = bf(rnd(M, W, seed=19))
    res = rnd(M, W, seed=20)
    y.backward(dy.float())
    yd = torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    K.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), eps, yd, mean, rstd)
    assert rel(yd.float(), y) < 4e-3
    y32 = torch.empty(M, W, device=DEV)
    K.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), eps, y32)
    assert rel(y32, y) < 1e-5
    dx, dxb = torch.empty(M, W, device=DEV), torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    dg, db = torch.zeros(W, device=DEV), torch.zeros(W, device=DEV)
    res2 = bf(rnd(M, W, seed=21))  # the side-branch residual term is a bf16 tensor
    K.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, g.to(DEV), dx, dx_bf16=dxb, res1=res.to(DEV), res2=res2.to(DEV),
                    dgamma=dg, dbeta=db)
    want = xr.grad + res + res2.float()
    assert rel(dx, want) < 1e-5
    assert rel(dxb.float(), want) < 4e-3
    assert rel(dg, gr.grad) < 1e-4 and rel(db, br.grad) < 1e-4
    # the atomic fallback of the dgamma / dbeta reduction (no workspace) gives the same sums
    dg2, db2 = torch.zeros(W, device=DEV), torch.zeros(W, device=DEV)
    K.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, g.to(DEV), dx, dgamma=dg2, dbeta=db2, workspace=False)
    assert rel(dg2, gr.grad) < 1e-4 and rel(db2, br.grad) < 1e-4
    # a bf16 input (side-branch value that only this LayerNorm consumes): forward and the residual-free backward
    xb = bf(x)
    xbr = xb.float().clone().requires_grad_(True)
    g2, b2 = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yb_ref = O.layer_norm(xbr, g2, b2, eps)
    yb_ref.backward(dy.float())
    yb = torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    mean_b, rstd_b = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    K.layernorm_fwd(xb.to(DEV), g.to(DEV), b.to(DEV), eps, yb, mean_b, rstd_b)
    assert rel(yb.float(), yb_ref) < 4e-3
    dxb3 = torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    dg3, db3 = torch.zeros(W, device=DEV), torch.zeros(W, device=DEV)
    K.layernorm_bwd(dy.to(DEV), xb.to(DEV), mean_b, rstd_b, g.to(DEV), None, dx_bf16=dxb3, dgamma=dg3, dbeta=db3)
    assert rel(dxb3.float(), xbr.grad) < 4e-3
    assert rel(dg3, g2.grad) < 1e-4 and rel(db3, b2.grad) < 1e-4
    # bf16-only output (no fp32 gradient tensor)
    dxb2 = torch.empty_like(dxb)
    K.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, g.to(DEV), None, dx_bf16=dxb2)
    assert rel(dxb2.float(), xr.grad) < 4e-3
    # the residual-stream gradient carried in bf16 (the space-time block's backward): bf16 res1, alone and with the bf16 side
    # branch, bf16-only output, dgamma / dbeta unchanged
    resb = bf(res)
    for r2 in (None, res2):
        dxb4 = torch.empty_like(dxb)
        dg4, db4 = torch.zeros(W, device=DEV), torch.zeros(W, device=DEV)
        K.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, g.to(DEV), None, dx_bf16=dxb4, res1=resb.to(DEV),
                        res2=r2.to(DEV) if r2 is not None else None, dgamma=dg4, dbeta=db4)
        want4 = xr.grad + resb.float() + (r2.float() if r2 is not None else 0)
        assert rel(dxb4.float(), want4) < 4e-3
        assert torch.equal(dg4, dg) and torch.equal(db4, db)


@pytest.mark.parametrize("W,period,q8", [(768, 7, False), (1280, 5, False), (256, 3, False), (768, 7, True)])
def test_layernorm_on_the_hybrid_stream(K, W, period, q8):
    """tvts_layernorm_{fwd,bwd}_cls: the stream is bf16, the rows r % period == 0 (a clip's CLS token) are carried in fp32 side
    arrays.  Those rows are normalised from the side array (their stream rows are stale garbage here) and receive their bf16
    rounding; in the backward their input comes from the side array, their residual-stream gradient from cls_res1 (fp32) and their
    result also goes to cls_dx in fp32; every other row behaves exactly like the plain bf16-stream kernels (same bits)."""
    Bc = 9
    M = Bc * period
    g, b = (1 + 0.1 * rnd(W, seed=17)).to(DEV), (0.1 * rnd(W, seed=18)).to(DEV)
    x_exact = rnd(M, W, seed=16) * 2 + 0.3                       # what the stream "is": fp32 on the CLS rows, bf16 elsewhere
    cls_rows = torch.arange(Bc) * period
    xs = bf(x_exact)
    xs[cls_rows] = 777.0                                          # stale stream rows: must never be read
    cls_x = x_exact[cls_rows].contiguous().to(DEV)
    truth = bf(x_exact).float()
    truth[cls_rows] = x_exact[cls_rows]
    xr = truth.clone().requires_grad_(True)
    gr, br = g.cpu().clone().requires_grad_(True), b.cpu().clone().requires_grad_(True)
    y_ref = O.layer_norm(xr, gr, br, 1e-5)
    dy = bf(rnd(M, W, seed=19))
    y_ref.backward(dy.float())
    x_dev = xs.to(DEV)
    y = torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    kw = {}
    if q8:
        kw = dict(q8=torch.empty(M, W, dtype=torch.uint8, device=DEV), row_scale=torch.empty(M, device=DEV))
    K.layernorm_fwd(x_dev, g, b, 1e-5, y, mean, rstd, cls_x=cls_x, cls_period=period, **kw)
    assert rel(y.float(), y_ref) < 4e-3
    assert torch.equal(x_dev[cls_rows.to(DEV)].view(torch.int16), cls_x.bfloat16().view(torch.int16))  # refreshed: rounded once
    # non-CLS rows: the bits of the plain bf16-input kernel
    y0 = torch.empty_like(y)
    m0, r0 = torch.empty_like(mean), torch.empty_like(rstd)
    K.layernorm_fwd(x_dev, g, b, 1e-5, y0, m0, r0)
    other = torch.ones(M, dtype=torch.bool); other[cls_rows] = False
    if not q8 and W % 8 == 0:  # (the fused-copy form is the 4-column kernel: same values, another summation order)
        assert torch.equal(y[other.to(DEV)].view(torch.int16), y0[other.to(DEV)].view(torch.int16))
    else:
        assert float((y[other.to(DEV)].float() - y0[other.to(DEV)].float()).abs().max()) <= 2.0 ** -7 * float(y0.float().abs().max())
    if q8:
        q0, rs0 = K.quantize_fp8_rows(y)
        assert torch.equal(q0, kw["q8"]) and torch.equal(rs0[:M], kw["row_scale"][:M])
        # the e4m3 bytes as the only output: same bytes, the CLS rows still refreshed
        x_dev2 = xs.to(DEV)
        kw2 = dict(q8=torch.full((M, W), 9, dtype=torch.uint8, device=DEV), row_scale=torch.empty(M, device=DEV))
        K.layernorm_fwd(x_dev2, g, b, 1e-5, None, torch.empty_like(mean), torch.empty_like(rstd), cls_x=cls_x, cls_period=period, **kw2)
        assert torch.equal(kw2["q8"], kw["q8"]) and torch.equal(kw2["row_scale"][:M], kw["row_scale"][:M])
        assert torch.equal(x_dev2.view(torch.int16), x_dev.view(torch.int16))
    # backward: stream gradient bf16, CLS rows of it fp32
    res1_exact = rnd(M, W, seed=20)
    res1 = bf(res1_exact); res1[cls_rows] = -555.0
    cls_res1 = res1_exact[cls_rows].contiguous().to(DEV)
    res2 = bf(rnd(M, W, seed=21))
    for form in ("res12", "res1", "none"):
        r1 = res1.to(DEV) if form != "none" else None
        r2 = res2.to(DEV) if form == "res12" else None
        dxb = torch.full((M, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        cls_dx = torch.full((Bc, W), float("nan"), device=DEV)
        dg, db = torch.zeros(W, device=DEV), torch.zeros(W, device=DEV)
        kwb = {}
        if q8:
            kwb = dict(q8=torch.empty(M, W, dtype=torch.uint8, device=DEV), row_scale=torch.empty(M, device=DEV))
        K.layernorm_bwd(dy.to(DEV), x_dev, mean, rstd, g, None, dx_bf16=dxb, res1=r1, res2=r2, dgamma=dg, dbeta=db,
                        cls_period=period, cls_x=cls_x, cls_res1=cls_res1 if form != "none" else None, cls_dx=cls_dx, **kwb)
        want = xr.grad.clone()
        if form != "none":
            r = bf(res1_exact).float(); r[cls_rows] = res1_exact[cls_rows]
            want = want + r
        if form == "res12":
            want = want + res2.float()
        assert rel(cls_dx, want[cls_rows]) < 1e-5, form
        assert rel(dxb.float(), want) < 4e-3, form
        assert torch.equal(dxb[cls_rows.to(DEV)].view(torch.int16), cls_dx.bfloat16().view(torch.int16)), form
        # (dgamma sums dy * xhat over every row in the plain launch, where a CLS row's xhat comes from its bf16-rounded stream row)
        assert rel(dg, gr.grad) < 1e-3 and rel(db, br.grad) < 1e-4, form
        if q8:
            q0, rs0 = K.quantize_fp8_rows(dxb)
            assert torch.equal(q0, kwb["q8"]) and torch.equal(rs0[:M], kwb["row_scale"][:M]), form


def test_layernorm_rows(K):
    """Gathered rows (ln_final on the EOT rows, sort-head norm on the caption rows) and the scatter in backward."""
    M, W, R = 40, 128, 6
    x = rnd(M, W, seed=21)
    rows = torch.tensor([3, 9, 10, 22, 39, 0], dtype=torch.int32)
    g, b = 1 + 0.1 * rnd(W, seed=22), 0.1 * rnd(W, seed=23)
    xr = x.clone().requires_grad_(True)
    y = O.layer_norm(xr[rows.long()], g, b, 1e-5)
    dy = rnd(R, W, seed=24)
    y.backward(dy)
    y32 = torch.empty(R, W, device=DEV)
    mean, rstd = torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    K.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, y32, mean, rstd, rows=rows.to(DEV))
    assert rel(y32, y) < 1e-5
    dx = torch.zeros(M, W, device=DEV)
    K.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, g.to(DEV), dx, rows=rows.to(DEV))
    assert rel(dx, xr.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ attention
def _ref_divided(qkv, heads, mode, T, n, dO):
    """Oracle divided attention on a given packed qkv (identity projections), + grads wrt qkv."""
    x = qkv.clone().requires_grad_(True)
    out = O.divided_attention_core(x, heads, mode, T, n)
    out.backward(dO)
    return out.detach(), x.grad


@pytest.mark.parametrize("tr", [True, False])
@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("mode,B,heads,T,n", [("time", 2, 2, 8, 5), ("space", 2, 2, 3, 21), ("space", 1, 3, 2, 98),
                                              ("time", 1, 4, 12, 3), ("space", 2, 1, 8, 49), ("time", 1, 2, 16, 4),
                                              ("space", 1, 2, 2, 76)])
def test_divided_attention(K, mode, B, heads, T, n, tr, dh):
    with K.options(attn_tr=tr):
        S, W = 1 + T * n, heads * dh
        qkv = bf(rnd(B, S, 3 * W, seed=25))
        dO = bf(rnd(B, S, W, seed=26))
        ref_out, ref_dqkv = _ref_divided(qkv.float(), heads, mode, T, n, dO.float())
        qd = qkv.reshape(B * S, 3 * W).to(DEV)
        out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * S, heads, device=DEV)
        K.attn_fwd(mode, qd, out, lse, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        K.attn_fwd("cls", qd, out, lse, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        assert rel(out.float().view(B, S, W), ref_out) < 8e-3, rel(out.float().view(B, S, W), ref_out)
        dOd = dO.reshape(B * S, W).to(DEV)
        delta = torch.empty(B * S, heads, device=DEV)
        dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
        acc = torch.zeros(B, heads, 3, dh, device=DEV)
        K.attn_delta(dOd, out, delta, rows=B * S, heads=heads, head_dim=dh)
        K.attn_bwd_dq(mode, qd, dOd, lse, delta, dqkv, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        K.attn_bwd_dq("cls", qd, dOd, lse, delta, dqkv, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        K.attn_bwd_dkv(mode, qd, dOd, lse, delta, dqkv, B=B, heads=heads, S=S, T=T, n=n, cls_acc=acc, head_dim=dh)
        K.attn_cls_finalize(acc, dqkv, B=B, heads=heads, S=S, head_dim=dh)
        got = dqkv.float().view(B, S, 3 * W).cpu()
        assert torch.isfinite(got).all()
        for nm, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
            assert rel(got[..., sl], ref_dqkv[..., sl]) < 2e-2, (nm, rel(got[..., sl], ref_dqkv[..., sl]))
            # CLS rows separately: they are the cross-group sums
            assert rel(got[:, 0, sl], ref_dqkv[:, 0, sl]) < 2e-2, (nm + "_cls", rel(got[:, 0, sl], ref_dqkv[:, 0, sl]))


@pytest.mark.parametrize("fused", [True, False, "atomic"])
@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("mode,B,heads,T,n", [("space", 2, 2, 3, 21), ("space", 1, 3, 2, 98), ("space", 2, 2, 4, 111),
                                              ("space", 1, 2, 2, 15), ("space", 1, 2, 2, 16), ("space", 1, 1, 2, 196),
                                              ("space", 2, 4, 3, 76), ("time", 2, 2, 8, 5), ("time", 1, 2, 16, 7),
                                              ("time", 2, 3, 12, 4), ("time", 2, 2, 8, 61)])
def test_attention_site_backward(K, mode, B, heads, T, n, dh, fused):
    """tvts_attn_bwd: the whole backward of one divided-attention site in one call (fused single-launch kernels where
    the group fits; the split delta / dQ / dK,dV passes otherwise), against autograd of the reference formulation.
    The CLS token's dK / dV / dQ shares: one partial per block added in order when the scratch has room for them (what the
    engine passes: two calls give bit-identical outputs), fp32 atomics with the minimal scratch ("atomic")."""
    parts = 1 if fused == "atomic" else max(T, -(-n // 28))
    fused = bool(fused)
    with K.options(attn_fused=fused):
        S, W = 1 + T * n, heads * dh
        qkv = bf(rnd(B, S, 3 * W, seed=35))
        dO = bf(rnd(B, S, W, seed=36))
        ref_out, ref_dqkv = _ref_divided(qkv.float(), heads, mode, T, n, dO.float())
        qd = qkv.reshape(B * S, 3 * W).to(DEV)
        out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * S, heads, device=DEV)
        K.attn_fwd(mode, qd, out, lse, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        K.attn_fwd("cls", qd, out, lse, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        dOd = dO.reshape(B * S, W).to(DEV)
        delta = torch.full((B * S, heads), float("nan"), device=DEV)
        dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
        acc = torch.full((B, heads, parts, 3, dh), float("nan"), device=DEV)
        K.attn_bwd(mode, qd, dOd, out, lse, delta, dqkv, B=B, heads=heads, S=S, T=T, n=n, cls_acc=acc, head_dim=dh)
        got = dqkv.float().view(B, S, 3 * W).cpu()
        assert torch.isfinite(got).all()
        if fused and parts > 1:  # ordered partials: a second call reproduces every bit
            dqkv2 = torch.full_like(dqkv, float("nan"))
            acc.fill_(float("nan"))
            K.attn_bwd(mode, qd, dOd, out, lse, delta, dqkv2, B=B, heads=heads, S=S, T=T, n=n, cls_acc=acc, head_dim=dh)
            assert torch.equal(dqkv2.view(torch.int16), dqkv.view(torch.int16))
        for nm, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
            assert rel(got[..., sl], ref_dqkv[..., sl]) < 2e-2, (nm, rel(got[..., sl], ref_dqkv[..., sl]))
            assert rel(got[:, 0, sl], ref_dqkv[:, 0, sl]) < 2e-2, (nm + "_cls", rel(got[:, 0, sl], ref_dqkv[:, 0, sl]))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("mode,B,heads,T,n", [("space", 2, 2, 3, 21), ("space", 1, 3, 2, 98), ("space", 2, 2, 4, 111),
                                              ("space", 1, 2, 2, 15), ("space", 1, 1, 2, 196), ("space", 2, 4, 3, 76),
                                              ("time", 2, 2, 8, 5), ("time", 1, 2, 16, 7), ("time", 2, 3, 12, 30),
                                              ("time", 1, 2, 8, 98)])
def test_attention_site_forward(K, mode, B, heads, T, n, dh, fused):
    """tvts_attn_fwd_divided: patch rows and the CLS row (merged from per-group partial softmax states) in one call."""
    with K.options(attn_fused=fused):
        S, W = 1 + T * n, heads * dh
        qkv = bf(rnd(B, S, 3 * W, seed=37))
        ref_out = O.divided_attention_core(qkv.float(), heads, mode, T, n)
        qd = qkv.reshape(B * S, 3 * W).to(DEV)
        out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.full((B * S, heads), float("nan"), device=DEV)
        ws = torch.full((B * heads * max(T, -(-n // 28)) * (dh + 2),), float("nan"), device=DEV)
        K.attn_fwd_divided(mode, qd, out, lse, ws, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh)
        got = out.float().view(B, S, W).cpu()
        assert torch.isfinite(got).all() and torch.isfinite(lse).all()
        assert rel(got, ref_out) < 8e-3, rel(got, ref_out)
        assert rel(got[:, 0], ref_out[:, 0]) < 8e-3, rel(got[:, 0], ref_out[:, 0])
        # lse (log2 domain) against the streaming kernels
        out2 = torch.empty_like(out)
        lse2 = torch.empty_like(lse)
        K.attn_fwd_divided(mode, qd, out2, lse2, ws, B=B, heads=heads, S=S, T=T, n=n, head_dim=dh, fused=False)
        assert float((lse - lse2).abs().max()) < 2e-3


def _ref_full(qkv, heads, causal, dO):
    B, S, W3 = qkv.shape
    W = W3 // 3
    dh = W // heads
    x = qkv.clone().requires_grad_(True)
    t = x.reshape(B, S, 3, heads, dh)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = (q * dh ** -0.5) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((S, S), float("-inf")).triu(1)
    o = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, S, W)
    o.backward(dO)
    return o.detach(), x.grad


@pytest.mark.parametrize("tr", [True, False])
@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("B,heads,S,causal", [(3, 2, 32, True), (2, 2, 77, True), (2, 2, 197, False), (1, 8, 789, False),
                                              (4, 1, 9, True)])
def test_full_attention(K, B, heads, S, causal, tr, dh):
    with K.options(attn_tr=tr):
        W = heads * dh
        qkv, dO = bf(rnd(B, S, 3 * W, seed=27)), bf(rnd(B, S, W, seed=28))
        ref_out, ref_d = _ref_full(qkv.float(), heads, causal, dO.float())
        qd, dOd = qkv.reshape(B * S, 3 * W).to(DEV), dO.reshape(B * S, W).to(DEV)
        out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse, delta = torch.empty(B * S, heads, device=DEV), torch.empty(B * S, heads, device=DEV)
        K.attn_fwd("full", qd, out, lse, B=B, heads=heads, S=S, causal=causal, head_dim=dh)
        assert rel(out.float().view(B, S, W), ref_out) < 8e-3
        dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.attn_delta(dOd, out, delta, rows=B * S, heads=heads, head_dim=dh)
        K.attn_bwd_dq("full", qd, dOd, lse, delta, dqkv, B=B, heads=heads, S=S, causal=causal, head_dim=dh)
        K.attn_bwd_dkv("full", qd, dOd, lse, delta, dqkv, B=B, heads=heads, S=S, causal=causal, head_dim=dh)
        got = dqkv.float().view(B, S, 3 * W).cpu()
        assert torch.isfinite(got).all()
        for nm, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
            assert rel(got[..., sl], ref_d[..., sl]) < 2e-2, (nm, rel(got[..., sl], ref_d[..., sl]))


@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("B,heads,S,causal", [(5, 2, 32, True), (3, 8, 32, False), (7, 3, 20, True), (4, 2, 16, True), (6, 1, 9, False),
                                              (3, 2, 2, True), (900, 8, 32, True)])
def test_short_sequence_attention(K, B, heads, S, causal, dh):
    """FULL attention over sequences of <= 32 tokens (the text tower's captions): the wave-per-(sequence, head) kernels behind
    tvts_attn_fwd / tvts_attn_bwd (D in registers, one launch) against autograd, and against the streaming kernels they replace.
    The last case has more groups than resident waves: the grid-stride walk with the next group's rows prefetched."""
    W = heads * dh
    qkv, dO = bf(rnd(B, S, 3 * W, seed=51)), bf(rnd(B, S, W, seed=52))
    qd, dOd = qkv.reshape(B * S, 3 * W).to(DEV), dO.reshape(B * S, W).to(DEV)

    def run():
        out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse, delta = torch.full((B * S, heads), float("nan"), device=DEV), torch.empty(B * S, heads, device=DEV)
        K.attn_fwd("full", qd, out, lse, B=B, heads=heads, S=S, causal=causal, head_dim=dh)
        dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.attn_bwd("full", qd, dOd, out, lse, delta, dqkv, B=B, heads=heads, S=S, causal=causal, head_dim=dh)
        return out.float().cpu(), lse.cpu(), dqkv.float().cpu()

    out, lse, got = run()
    with K.options(attn_fused=False):
        out_s, lse_s, got_s = run()
    assert torch.isfinite(out).all() and torch.isfinite(lse).all() and torch.isfinite(got).all()
    assert rel(out, out_s) < 4e-3 and (lse - lse_s).abs().max() < 1e-3 and rel(got, got_s) < 8e-3
    if B <= 16:
        ref_out, ref_d = _ref_full(qkv.float(), heads, causal, dO.float())
        assert rel(out.view(B, S, W), ref_out) < 8e-3
        g3 = got.view(B, S, 3 * W)
        for nm, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
            assert rel(g3[..., sl], ref_d[..., sl]) < 2e-2, (nm, rel(g3[..., sl], ref_d[..., sl]))


@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("B,heads,S,nq", [(3, 2, 200, 4), (2, 8, 789, 4), (2, 1, 37, 1), (1, 2, 130, 16)])
def test_tail_query_attention(K, B, heads, S, nq, dh):
    """FULL attention whose only queries are the last nq tokens of every sequence (the sort head's last block): output rows and
    dQ of the query rows, dK / dV of every row, against autograd of the same restriction."""
    W = heads * dh
    qkv, dO = bf(rnd(B, S, 3 * W, seed=41)), bf(rnd(B, S, W, seed=42))
    dO[:, :S - nq] = 0  # only the query rows carry an upstream gradient
    ref_out, ref_d = _ref_full(qkv.float(), heads, False, dO.float())
    qd, dOd = qkv.reshape(B * S, 3 * W).to(DEV), dO.reshape(B * S, W).to(DEV)
    out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse, delta = torch.full((B * S, heads), float("nan"), device=DEV), torch.full((B * S, heads), float("nan"), device=DEV)
    K.attn_fwd_tail(qd, out, lse, B=B, heads=heads, S=S, nq=nq, head_dim=dh)
    o = out.float().view(B, S, W).cpu()
    assert torch.isnan(o[:, :S - nq]).all() and torch.isfinite(o[:, S - nq:]).all()   # only the query rows are written
    assert rel(o[:, S - nq:], ref_out[:, S - nq:]) < 8e-3
    dqkv = torch.zeros(B * S, 3 * W, dtype=torch.bfloat16, device=DEV)
    dOd2 = dOd.clone(); dOd2.view(B, S, W)[:, :S - nq] = float("nan")                 # ... and only they are read
    K.attn_bwd_tail(qd, dOd2, out, lse, delta, dqkv, B=B, heads=heads, S=S, nq=nq, head_dim=dh)
    got = dqkv.float().view(B, S, 3 * W).cpu()
    assert torch.isfinite(got).all()
    assert (got[:, :S - nq, :W] == 0).all()
    for nm, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
        assert rel(got[..., sl], ref_d[..., sl]) < 2e-2, (nm, rel(got[..., sl], ref_d[..., sl]))


@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("B,heads,S", [(5, 2, 32), (3, 8, 77), (2, 1, 9)])
def test_row_query_attention(K, B, heads, S, dh):
    """One query per sequence at a ragged position (the EOT token of the text tower's last block) under the causal mask."""
    W = heads * dh
    qkv, dO = bf(rnd(B, S, 3 * W, seed=43)), bf(rnd(B, S, W, seed=44))
    qpos = torch.tensor([(7 * b + 3) % S for b in range(B)], dtype=torch.int32)
    qpos[0] = S - 1
    keep = torch.zeros(B, S, 1)
    keep[torch.arange(B), qpos.long()] = 1
    dO = dO * keep.to(dO.dtype)
    ref_out, ref_d = _ref_full(qkv.float(), heads, True, dO.float())
    qd, dOd = qkv.reshape(B * S, 3 * W).to(DEV), dO.reshape(B * S, W).to(DEV)
    out = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse, delta = torch.full((B * S, heads), float("nan"), device=DEV), torch.full((B * S, heads), float("nan"), device=DEV)
    K.attn_fwd_rowq(qd, qpos.to(DEV), out, lse, B=B, heads=heads, S=S, head_dim=dh)
    o = out.float().view(B, S, W).cpu()
    idx = (torch.arange(B), qpos.long())
    assert torch.isfinite(o[idx]).all() and int(torch.isfinite(o).all(-1).sum()) == B
    assert rel(o[idx], ref_out[idx]) < 8e-3
    dqkv = torch.full((B * S, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
    K.attn_bwd_rowq(qd, qpos.to(DEV), dOd, torch.nan_to_num(out), lse, delta, dqkv, B=B, heads=heads, S=S, head_dim=dh)
    got = dqkv.float().view(B, S, 3 * W).cpu()
    assert torch.isfinite(got).all()
    for nm, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
        assert rel(got[..., sl], ref_d[..., sl]) < 2e-2, (nm, rel(got[..., sl], ref_d[..., sl]))


def test_attention_softmax_spike(K):
    """Online-softmax rescale across key tiles: one key far above the rest in a late tile."""
    B, heads, S = 1, 1, 200
    qkv = bf(rnd(B, S, 192, seed=29))
    qkv[0, 5, 0:64] = 6.0
    qkv[0, 150, 64:128] = 6.0   # q5 . k150 = 2304/8 = 288 >> others
    dO = bf(rnd(B, S, 64, seed=30))
    ref_out, _ = _ref_full(qkv.float(), heads, False, dO.float())
    out = torch.empty(S, 64, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(S, 1, device=DEV)
    K.attn_fwd("full", qkv.reshape(S, 192).to(DEV), out, lse, B=1, heads=1, S=S)
    assert rel(out.float().view(1, S, 64), ref_out) < 8e-3
    assert rel(out.float()[5], ref_out[0, 5]) < 8e-3


# ------------------------------------------------------------------------------------------------ embed kernels
def test_patch_embed_and_assemble(K):
    from tvts_amd import arch as A
    arch = A.small_arch()
    oarch = O.tiny_arch(**{k: arch[k] for k in ("image", "patch", "width", "heads", "layers", "embed", "mask_ratio")})
    B, T = 2, 3
    P = O.synth_params(oarch, seed=31)
    batch = O.synth_batch(oarch, B=B, T=T, seed=32)
    n, W, p = batch["keep_ind"].shape[1], arch["width"], arch["patch"]
    keep = batch["keep_ind"].to(torch.int32).to(DEV)
    cols = torch.empty(B * T * n, 3 * p * p, dtype=torch.bfloat16, device=DEV)
    K.patch_gather(batch["video"].to(DEV), keep, cols, B=B, T=T, n=n, img=arch["image"], patch=p)
    g = arch["image"] // p
    pix = batch["video"].reshape(B, T, 3, g, p, g, p).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, g * g, 3 * p * p)
    ref_cols = torch.gather(pix, 2, batch["keep_ind"][:, None, :, None].expand(B, T, n, 3 * p * p)).reshape(-1, 3 * p * p)
    assert torch.equal(cols.float().cpu(), bf(ref_cols).float())
    # assemble + ln_pre against the oracle's token embedding (with a bf16 conv GEMM in between)
    wconv = P["video_model.conv1.weight"].reshape(W, -1)
    pe = torch.empty(B * T * n, W, device=DEV)
    K.gemm_nt(cols, bf(wconv).to(DEV), pe)
    tok = torch.empty(B * (1 + T * n), W, device=DEV)
    K.vit_assemble(pe, P["video_model.class_embedding"].to(DEV), P["video_model.positional_embedding"].to(DEV),
                   P["video_model.temporal_embedding"].to(DEV), keep, tok, B=B, T=T, n=n)
    x = torch.empty_like(tok)
    K.layernorm_fwd(tok, P["video_model.ln_pre.weight"].to(DEV), P["video_model.ln_pre.bias"].to(DEV), 1e-5, x)
    ref = O.video_embed_tokens(P, batch["video"], batch["keep_ind"], oarch).reshape(-1, W)
    assert rel(x, ref) < 5e-3
    # backward scatter
    dtok = rnd(B * (1 + T * n), W, seed=33)
    tokr = {k: P[k].clone().requires_grad_(True) for k in ("video_model.class_embedding", "video_model.positional_embedding",
                                                           "video_model.temporal_embedding")}
    S = 1 + T * n
    pos = tokr["video_model.positional_embedding"]
    patch_in = torch.zeros(B, T, n, W, requires_grad=True)
    tk = patch_in + pos[1:][batch["keep_ind"]][:, None] + tokr["video_model.temporal_embedding"][:T][None, :, None]
    full = torch.cat([(tokr["video_model.class_embedding"] + pos[0]).expand(B, 1, W), tk.reshape(B, T * n, W)], 1)
    full.backward(dtok.view(B, S, W))
    dpatch = torch.empty(B * T * n, W, dtype=torch.bfloat16, device=DEV)
    dcls, dpos, dtmp = torch.zeros(W, device=DEV), torch.zeros(g * g + 1, W, device=DEV), torch.zeros(12, W, device=DEV)
    K.vit_assemble_bwd(dtok.to(DEV), keep, dpatch, dcls, dpos, dtmp, B=B, T=T, n=n)
    assert rel(dpatch.float(), patch_in.grad.reshape(-1, W)) < 4e-3
    assert rel(dcls, tokr["video_model.class_embedding"].grad) < 1e-5
    assert rel(dpos, pos.grad) < 1e-5
    assert rel(dtmp, tokr["video_model.temporal_embedding"].grad) < 1e-5
    # every sum is ordered (per-block partials, positional rows gathered in clip order): the same bits on every run
    dcls2, dpos2, dtmp2 = torch.zeros(W, device=DEV), torch.zeros(g * g + 1, W, device=DEV), torch.zeros(12, W, device=DEV)
    K.vit_assemble_bwd(dtok.to(DEV), keep, dpatch, dcls2, dpos2, dtmp2, B=B, T=T, n=n)
    assert torch.equal(dcls, dcls2) and torch.equal(dpos, dpos2) and torch.equal(dtmp, dtmp2)


def test_text_embed_mean_sort_assemble(K):
    N, L, Wt, V, ctx = 6, 7, 128, 50, 12
    ids = torch.randint(0, V, (N, ctx), generator=torch.Generator().manual_seed(34), dtype=torch.int32)
    emb, pos = rnd(V, Wt, seed=35), rnd(ctx, Wt, seed=36)
    x = torch.empty(N * L, Wt, device=DEV)
    K.text_embed(ids.to(DEV), emb.to(DEV), pos.to(DEV), x, N=N, L=L)
    ref = emb[ids[:, :L].long()] + pos[:L]
    assert rel(x, ref.reshape(-1, Wt)) < 1e-6
    dx = rnd(N * L, Wt, seed=37)
    demb, dpos = torch.zeros(V, Wt, device=DEV), torch.zeros(ctx, Wt, device=DEV)
    K.text_embed_bwd(dx.to(DEV), ids.to(DEV), demb, dpos, N=N, L=L)
    e2 = emb.clone().requires_grad_(True); p2 = pos.clone().requires_grad_(True)
    (e2[ids[:, :L].long()] + p2[:L]).backward(dx.view(N, L, Wt))
    assert rel(demb, e2.grad) < 1e-5 and rel(dpos, p2.grad) < 1e-5
    # ... and as ordered sums over the rows sorted by token id (what the engines pass): the same values, the same bits every time
    order, seg = K.token_sort(ids[:, :L])
    assert sorted(order.tolist()) == list(range(N * L)) and seg.numel() == N * L + 1 and int(seg[-1]) == N * L
    flat = ids[:, :L].reshape(-1)
    assert all(flat[order[i]] < flat[order[i + 1]] or (flat[order[i]] == flat[order[i + 1]] and order[i] < order[i + 1]) for i in range(N * L - 1))
    outs = []
    for _ in range(2):
        demb2, dpos2 = torch.zeros(V, Wt, device=DEV), torch.zeros(ctx, Wt, device=DEV)
        K.text_embed_bwd(dx.to(DEV), ids.to(DEV), demb2, dpos2, N=N, L=L, tok_sort=(order.to(DEV), seg.to(DEV)))
        assert rel(demb2, e2.grad) < 1e-6 and rel(dpos2, p2.grad) < 1e-6
        outs.append((demb2, dpos2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # long runs (a token every caption shares: its start and end token) are listed first and summed by a block per 64 columns
    N2, L2, V2, W2 = 300, 5, 40, 192
    ids2 = torch.randint(3, V2, (N2, L2), generator=torch.Generator().manual_seed(51), dtype=torch.int32)
    ids2[:, 0], ids2[:, -1] = 1, 2
    ids2[:70, 2] = 7
    dx2 = rnd(N2 * L2, W2, seed=52)
    order2, seg2 = K.token_sort(ids2)
    flat2 = ids2.reshape(-1)
    lens2 = torch.diff(seg2[:int((seg2 < N2 * L2).sum()) + 1].long())
    assert sorted(order2.tolist()) == list(range(N2 * L2)) and lens2[:2].tolist() == [300, 300] and int(lens2[2]) > 64 and int(lens2[3:].max()) <= 64
    for s0, s1 in zip(seg2[:-1].tolist(), seg2[1:].tolist()):
        run = order2[s0:s1]
        assert len(set(flat2[run.long()].tolist())) <= 1 and run.tolist() == sorted(run.tolist())
    e3 = torch.zeros(V2, W2, requires_grad=True)
    e3[ids2.long()].backward(dx2.view(N2, L2, W2))
    res = []
    for _ in range(2):
        demb3, dpos3 = torch.zeros(V2, W2, device=DEV), torch.zeros(L2, W2, device=DEV)
        K.text_embed_bwd(dx2.to(DEV), ids2.to(DEV), demb3, dpos3, N=N2, L=L2, tok_sort=(order2.to(DEV), seg2.to(DEV)))
        assert rel(demb3, e3.grad) < 1e-6 and rel(dpos3, dx2.view(N2, L2, W2).sum(0)) < 1e-6
        res.append(demb3)
    assert torch.equal(res[0], res[1])
    # caption mean (clip-major) and its backward
    NT, B, E = 4, 3, 128
    t = rnd(NT * B, E, seed=38)
    mean, before = torch.empty(B, E, device=DEV), torch.empty(B, NT, E, device=DEV)
    K.text_mean(t.to(DEV), mean, before, NT=NT, B=B)
    assert rel(mean, t.view(NT, B, E).mean(0)) < 1e-6 and rel(before, t.view(NT, B, E).permute(1, 0, 2)) < 1e-7
    dmean = rnd(B, E, seed=39)
    dt = torch.empty(NT * B, E, device=DEV)
    K.text_mean_bwd(dmean.to(DEV), dt, NT=NT, B=B)
    assert rel(dt, (dmean / NT).repeat(NT, 1)) < 1e-6
    # sort-head assemble / backward
    S = 5
    tok, txt, ty = rnd(B * S, E, seed=40), rnd(B, NT, E, seed=41), rnd(2, E, seed=42)
    xs = torch.empty(B * (S + NT), E, device=DEV)
    K.sort_assemble(tok.to(DEV), txt.to(DEV), ty.to(DEV), xs, B=B, S=S, off=0, Sv=S, NT=NT)
    ref = torch.cat([tok.view(B, S, E) + ty[0], txt + ty[1]], 1)
    assert rel(xs, ref.reshape(-1, E)) < 1e-6
    dxs, dvid = rnd(B * (S + NT), E, seed=43), rnd(B, E, seed=44)
    dout = torch.empty(B * S, E, dtype=torch.bfloat16, device=DEV)
    dty = torch.zeros(2, E, device=DEV)
    K.sort_assemble_bwd(dxs.to(DEV), dvid.to(DEV), dout, dty, B=B, S=S, off=0, Sv=S, NT=NT)
    d3 = dxs.view(B, S + NT, E)
    rd = d3[:, :S].clone(); rd[:, 0] += dvid
    assert rel(dout.float(), rd.reshape(-1, E)) < 4e-3
    assert rel(dty, torch.stack([d3[:, :S].sum((0, 1)), d3[:, S:].sum((0, 1))])) < 1e-5
    K.sort_assemble_bwd(None, dvid.to(DEV), dout, None, B=B, S=S, off=0, Sv=S, NT=NT)
    rd = torch.zeros(B, S, E); rd[:, 0] = dvid
    assert rel(dout.float(), rd.reshape(-1, E)) < 4e-3


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("G", [6, 96, 200])
def test_losses(K, G):
    from tvts_amd.engine import LossHead
    E = 128
    v, t = rnd(G, E, seed=45), rnd(G, E, seed=46)
    vr, tr_ = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    ref = O.norm_softmax_loss(O.sim_matrix(vr, tr_))
    ref.backward()
    head = LossHead(torch.device(DEV))
    loss, dv, dt = head.contrastive(v.to(DEV), t.to(DEV))
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1, abs(float(ref)))
    assert rel(dv, vr.grad) < 1e-4 and rel(dt, tr_.grad) < 1e-4
    pred = rnd(G, 4, seed=47).requires_grad_(True)
    lab = torch.arange(4).repeat(G // 4 + 1)[:G]
    r2 = O.sorting_ce(pred, lab)
    r2.backward()
    l2, dp = head.sorting(pred.detach().to(DEV), lab.to(torch.int32).to(DEV))
    assert abs(float(l2) - float(r2)) < 1e-5 and rel(dp, pred.grad) < 1e-5


@pytest.mark.parametrize("G,E", [(1536, 512), (4096, 512), (1000, 1024), (1537, 64)])
def test_losses_at_gathered_batch_sizes(K, G, E):
    """The contrastive head at the gathered sizes of 8 GPUs (G = 8 x 192 = 1536) and beyond: the G x G x E similarity and its
    two backward products run on the fp32 MFMA kernel, the column log-sum-exp on the coalesced kernel.  Reference = the
    same formulas in float64 torch on the GPU."""
    from tvts_amd.engine import LossHead
    g = torch.Generator(device=DEV).manual_seed(G)
    v, t = torch.randn(G, E, generator=g, device=DEV), torch.randn(G, E, generator=g, device=DEV)
    vr, tr_ = v.double().requires_grad_(True), t.double().requires_grad_(True)
    x = torch.nn.functional.normalize(vr, dim=1) @ torch.nn.functional.normalize(tr_, dim=1).t() / 0.05
    ref = -(torch.log_softmax(x, 1).diagonal().mean() + torch.log_softmax(x.t(), 1).diagonal().mean())
    ref.backward()
    head = LossHead(torch.device(DEV))
    loss, dv, dt = head.contrastive(v, t)
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1, abs(float(ref))), (float(loss), float(ref))
    assert rel(dv, vr.grad) < 1e-4 and rel(dt, tr_.grad) < 1e-4, (rel(dv, vr.grad), rel(dt, tr_.grad))


@pytest.mark.parametrize("M,N,K_", [(64, 64, 16), (100, 70, 50), (1536, 1536, 512), (333, 512, 1000), (32, 32, 16), (65, 33, 17)])
def test_gemm_small_mfma_tile_all_stride_patterns(K, M, N, K_):
    """tvts_gemm_small_f32 on the fp32 MFMA tile (M, N >= 32, K >= 16): row-major and transposed views of both operands,
    alpha, bias, accumulate, ragged edges.  fp32 MFMA is an exact fp32 fma chain: float64 reference to 1e-5."""
    a, b = rnd(M, K_, seed=80).to(DEV), rnd(K_, N, seed=81).to(DEV)
    at, bt = a.t().contiguous(), b.t().contiguous()  # [K,M], [N,K]
    bias = rnd(N, seed=82).to(DEV)
    ref = a.double() @ b.double()
    for A_, sa in ((a, (K_, 1)), (at, (1, M))):
        for B_, sb in ((b, (N, 1)), (bt, (1, K_))):
            out = torch.full((M, N + 3), float("nan"), device=DEV)[:, :N]
            K.gemm_small(A_, B_, out, M=M, N=N, K=K_, sa=sa, sb=sb, alpha=0.5, bias=bias)
            assert rel(out, 0.5 * ref + bias.double()) < 1e-5, (sa, sb, rel(out, 0.5 * ref + bias.double()))
            K.gemm_small(A_, B_, out, M=M, N=N, K=K_, sa=sa, sb=sb, accumulate=True)
            assert rel(out, 1.5 * ref + bias.double()) < 1e-5


# ------------------------------------------------------------------------------------------------ optimizer
def test_adamw_and_shadows(K):
    n = 4096 * 3
    p, g = rnd(n, seed=48), rnd(n, seed=49) * 0.1
    m, v = torch.zeros(n), torch.zeros(n)
    groups = torch.tensor([0] * 4 + [3] * 4 + [255] * 4, dtype=torch.uint8)
    lr4, wd4 = [1e-2, 0, 0, 1e-3], [0.05, 0, 0, 0.0]
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    step_dev = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in (1, 2, 3):
        gs = g * step
        if step < 3:
            K.adamw_hf(pd, gs.to(DEV), md, vd, sh, groups.to(DEV), lr4, wd4, step, grad_scale=0.5)
        else:  # device-side step counter (graph-replay form)
            step_dev.fill_(3)
            K.adamw_hf(pd, gs.to(DEV), md, vd, sh, groups.to(DEV), lr4, wd4, 0, grad_scale=0.5, step_dev=step_dev)
        for lo, hi, gi in ((0, 4096, 0), (4096, 8192, 3)):
            O.hf_adamw_step(pr[lo:hi], 0.5 * gs[lo:hi], mr[lo:hi], vr[lo:hi], step, lr4[gi], wd4[gi])
    assert rel(pd, pr) < 1e-6 and rel(md, mr) < 1e-6 and rel(vd, vr) < 1e-6
    assert torch.equal(pd[8192:].cpu(), p[8192:])  # frozen chunks untouched
    assert torch.equal(sh[:8192].float().cpu(), bf(pd[:8192].cpu()).float())
    c = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    K.cast_f32_bf16(pd, c)
    assert torch.equal(c.float().cpu(), bf(pd.cpu()).float())


def test_param_store_shadows(K):
    from tvts_amd import arch as A
    from tvts_amd.engine import ParamStore
    st = ParamStore(A.small_arch(), torch.device(DEV))
    st.flat.copy_(torch.randn(st.total, generator=torch.Generator().manual_seed(50)))
    st.refresh_shadows()
    for name in ("video_model.conv1.weight", "video_model.proj", "text_model.resblocks.1.mlp.c_fc.weight",
                 "pred_model.blocks.0.attn.qkv.weight"):
        w = st.p(name).reshape(st.shapes[name][0], -1)
        assert torch.equal(st.w(name).float(), w.to(torch.bfloat16).float())
        assert torch.equal(st.wt(name).float(), w.t().to(torch.bfloat16).float())


def test_patch_gather_14x14_padded_k(K):
    """H/14 patches: 3*14*14 = 588 columns in conv-weight order, zero-padded to 640 (bit-exact bf16 rounding of the
    pixels), the padded weight copy, and the padded wgrad folded back into a [W, 588] gradient."""
    B, T, img, p, n, W = 2, 3, 56, 14, 5, 64
    g = img // p
    gen = torch.Generator().manual_seed(44)
    video = torch.randn(B, T, 3, img, img, generator=gen)
    keep_ind = torch.stack([torch.randperm(g * g, generator=gen)[:n].sort().values for _ in range(B)])
    keep = keep_ind.to(torch.int32).to(DEV)
    Kc, Kp = 3 * p * p, 640
    cols = torch.full((B * T * n, Kp), 7.0, dtype=torch.bfloat16, device=DEV)
    K.patch_gather(video.to(DEV), keep, cols, B=B, T=T, n=n, img=img, patch=p)
    pix = video.reshape(B, T, 3, g, p, g, p).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, g * g, Kc)
    ref_cols = torch.gather(pix, 2, keep_ind[:, None, :, None].expand(B, T, n, Kc)).reshape(-1, Kc)
    assert torch.equal(cols[:, :Kc].float().cpu(), bf(ref_cols).float())
    assert float(cols[:, Kc:].float().abs().max()) == 0.0
    w = bf(rnd(W, Kc, seed=45)).to(DEV)
    wp = torch.full((W, Kp), 3.0, dtype=torch.bfloat16, device=DEV)
    K.pad_rows_bf16(w, wp)
    assert torch.equal(wp[:, :Kc], w) and float(wp[:, Kc:].float().abs().max()) == 0.0
    pe = torch.empty(B * T * n, W, device=DEV)
    K.gemm_nt(cols, wp, pe)
    assert rel(pe, bf(ref_cols).float() @ w.float().cpu().t()) < 1e-5
    dst = rnd(W, Kc, seed=46).to(DEV)
    src = rnd(W, Kp, seed=47).to(DEV)
    want = dst.cpu() + src.cpu()[:, :Kc]
    K.add_rows_f32(dst, src)
    assert torch.equal(dst.cpu(), want)


def _e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fp8_quantisation(K, dtype):
    """per-tensor e4m3 (OCP) quantisation: bit patterns of torch.float8_e4m3fn on x * 448 / amax."""
    x = (rnd(300, 256, seed=51) * 3).to(dtype)
    x[5, 7] = -17.0
    q, scale = K.quantize_fp8(x.to(DEV))
    s = float(x.float().abs().max()) / 448.0
    assert abs(float(scale) - s) < 1e-7 * s
    want = _e4m3(x.float() * (1.0 / torch.tensor(s, dtype=torch.float32)))
    got = q.cpu().view(torch.float8_e4m3fn)
    diff = (got.float() - want.float()).abs()
    # identical except where x / scale sits within rounding of a tie (the kernel multiplies by 1 / scale in fp32)
    assert float((diff > 0).float().mean()) < 2e-3 and float((diff / want.float().abs().clamp_min(1e-3)).max()) <= 0.13
    # a strided view (columns of a wider matrix)
    wide = torch.zeros(300, 512, dtype=dtype, device=DEV)
    wide[:, 128:384] = x.to(DEV)
    q2, s2 = K.quantize_fp8(wide[:, 128:384])
    assert torch.equal(q2, q) and torch.equal(s2, scale)


@pytest.mark.parametrize("M,N,K_", [(256, 256, 128), (1000, 768, 768), (777, 1280, 1280), (9420, 3072, 768)])
def test_gemm_nt_fp8(K, M, N, K_):
    """out = sa*sb * (A8 B8^T) + bias: exact products of the e4m3 values, fp32 accumulation."""
    a, b = rnd(M, K_, seed=52), rnd(N, K_, seed=53) * K_ ** -0.5
    a8, sa = K.quantize_fp8(a.to(DEV))
    b8, sb = K.quantize_fp8(b.to(DEV))
    bias = rnd(N, seed=54).to(DEV)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    K.gemm_nt_fp8(a8, sa, b8, sb, out, bias=bias)
    ad, bd = a8.cpu().view(torch.float8_e4m3fn).float(), b8.cpu().view(torch.float8_e4m3fn).float()
    ref = (ad.double() @ bd.double().t()) * (float(sa) * float(sb)) + bias.cpu().double()
    assert rel(out, ref) < 5e-5, rel(out, ref)  # fp32 accumulation inside the matrix pipe
    # against the unquantised product: the quantisation error of two e4m3 tensors
    assert rel(out, a.double() @ b.double().t() + bias.cpu().double()) < 0.06
    outb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    res = rnd(M, N, seed=55).to(DEV)
    K.gemm_nt_fp8(a8, sa, b8, sb, outb)
    assert rel(outb.float(), ref - bias.cpu().double()) < 4e-3
    K.gemm_nt_fp8(a8, sa, b8, sb, out, bias=bias, residual=res)
    assert rel(out, ref + res.cpu().double()) < 5e-5


@pytest.mark.parametrize("M,Na,Nb", [(128, 256, 256), (1000, 512, 256), (4097, 1280, 640), (9420, 768, 2304), (58416 // 4, 1280, 5120), (333, 272, 48)])
def test_gemm_tn_fp8(K, M, Na, Nb):
    """e4m3 weight gradient (BASELINE config 5): out (+)= sp * sq * P8^T Q8 through ds_read_b64_tr_b8 fragments and the K = 128 scaled
    MFMA -- exact products of the e4m3 values with fp32 accumulation, token ranges that end inside a 128-row stage, ragged tiles,
    accumulate / overwrite, with and without the split workspace, and the same bits launch after launch."""
    p, q = rnd(M, Na, seed=61), rnd(M, Nb, seed=62) * 0.5
    p8, sp = K.quantize_fp8(p.to(DEV))
    q8, sq = K.quantize_fp8(q.to(DEV))
    pd, qd = p8.cpu().view(torch.float8_e4m3fn).float().double(), q8.cpu().view(torch.float8_e4m3fn).float().double()
    ref = (pd.t() @ qd) * (float(sp) * float(sq))
    out = torch.full((Na, Nb), float("nan"), dtype=torch.float32, device=DEV)
    K.gemm_tn_fp8(p8, sp, q8, sq, out, accumulate=False)
    assert rel(out, ref) < 5e-5, rel(out, ref)
    assert rel(out, p.double().t() @ q.double()) < 0.06   # against the unquantised product: two e4m3 tensors' rounding
    out2 = torch.full((Na, Nb), float("nan"), dtype=torch.float32, device=DEV)
    K.gemm_tn_fp8(p8, sp, q8, sq, out2, accumulate=False)
    assert torch.equal(out, out2)
    base = rnd(Na, Nb, seed=63).to(DEV)
    out3 = base.clone()
    K.gemm_tn_fp8(p8, sp, q8, sq, out3, accumulate=True)
    assert rel(out3, ref + base.cpu().double()) < 5e-5
    out4 = base.clone()
    K.gemm_tn_fp8(p8, sp, q8, sq, out4, accumulate=True, workspace=False)   # one range, read-modify-write epilogue
    assert rel(out4, ref + base.cpu().double()) < 5e-5
    cs = torch.ones(Na, device=DEV)   # bias gradient from the same bytes: scale_p * column sums of the e4m3 values
    K.gemm_tn_fp8(p8, sp, q8, sq, out2, accumulate=False, colsum=cs)
    assert torch.equal(out, out2) and rel(cs, 1 + pd.sum(0) * float(sp)) < 2e-5, rel(cs, 1 + pd.sum(0) * float(sp))
    cs2 = torch.ones(Na, device=DEV)
    K.gemm_tn_fp8(p8, sp, q8, sq, out2, accumulate=False, colsum=cs2, workspace=False)
    assert rel(cs2, 1 + pd.sum(0) * float(sp)) < 2e-5
    K.gemm_tn_fp8(p8, sp, q8, sq, out4, accumulate=False, splits=3) if M >= 3072 else None
    if M >= 3072:
        assert rel(out4, ref) < 5e-5
    # column slices of wider byte matrices (leading dimension != width)
    if Na >= 512:
        K.gemm_tn_fp8(p8[:, 256:512], sp, q8[:, :Nb // 2 // 16 * 16], sq, out[:256, :Nb // 2 // 16 * 16], accumulate=False)
        assert rel(out[:256, :Nb // 2 // 16 * 16], ref[256:512, :Nb // 2 // 16 * 16]) < 5e-5


def test_gemm_nt_fp8_main_loops_agree_and_grid_limit(K):
    """the K = 128 scaled-MFMA main loop and the 16x16x32 fp8 loop accumulate the same e4m3 products in fp32: identical bits on a
    shape with ragged row tiles, every epilogue form; a persistent grid limited to 64 CUs (TVTS_GEMM_CUS in the call's opts) changes
    nothing, for the fp8 and for the bf16 256x256 kernel."""
    M, N, K_ = 9000, 1280, 640
    a, b = rnd(M, K_, seed=81), rnd(N, K_, seed=82) * K_ ** -0.5
    a8, rs = K.quantize_fp8_rows(a.bfloat16().to(DEV))
    b8, sb = K.quantize_fp8(b.to(DEV))
    bias, res = rnd(N, seed=83).to(DEV), rnd(M, N, seed=84).to(DEV)
    h = rnd(M, N, seed=85).bfloat16().to(DEV)

    def run():
        o1 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); pre = torch.empty_like(o1)
        K.gemm_nt_fp8(a8, rs, b8, sb, o1, bias=bias, act="gelu", preact=pre)
        o2 = torch.empty(M, N, dtype=torch.float32, device=DEV)
        K.gemm_nt_fp8(a8, rs, b8, sb, o2, bias=bias, residual=res)
        o3 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        K.gemm_nt_fp8(a8, rs, b8, sb, o3, gate_h=h, gate_act="gelu")
        return o1, pre, o2, o3

    if True:
        ref = run()
        with K.options(fp8_k32=True):
            old = run()[:3]            # the 16x16x32 loop has no gated form
        for x, y in zip(ref[:3], old):
            assert torch.equal(x, y)
        with K.options(nt_cus=64):
            for x, y in zip(ref, run()):
                assert torch.equal(x, y)
        ab, bb = a.bfloat16().to(DEV), b.bfloat16().to(DEV)
        o64 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); K.gemm_nt(ab, bb, o64, bias=bias, tile=256, cus=64)
        o256 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); K.gemm_nt(ab, bb, o256, bias=bias, tile=256)
        assert torch.equal(o64, o256)
        # the gated fp8 form against its definition
        ad, bd = a8.cpu().view(torch.float8_e4m3fn).float(), b8.cpu().view(torch.float8_e4m3fn).float()
        z = (ad.double() @ bd.double().t()) * rs.cpu().double()[:M, None] * float(sb)
        hf = h.float().cpu().double()
        gate = 0.5 * (1 + torch.erf(hf / 2 ** 0.5)) + hf * torch.exp(-0.5 * hf * hf) / (2 * torch.pi) ** 0.5
        assert rel(ref[3].float(), z * gate) < 4e-3


@pytest.mark.parametrize("W,form", [(768, "res1"), (768, "res12"), (1280, "res1"), (1280, "res12"), (1280, "xbf16"), (640, "plain"), (256, "xbf16")])
def test_layernorm_bwd_fp8_output(K, W, form):
    """LayerNorm backward with the fused e4m3 copy of dx_bf16: dx / dx_bf16 / dgamma / dbeta identical to the plain kernel, the fp8
    bytes and per-row scales identical to quantize_fp8_rows(dx_bf16)."""
    M = 517
    xdt = torch.bfloat16 if form == "xbf16" else torch.float32
    x = (rnd(M, W, seed=90) * 2 + 0.3).to(xdt).to(DEV)
    g = (1 + 0.1 * rnd(W, seed=91)).to(DEV)
    b = (0.1 * rnd(W, seed=92)).to(DEV)
    dy = (rnd(M, W, seed=93) * torch.logspace(-4, 1, M)[:, None]).bfloat16().to(DEV)
    dy[3] = 0
    res1 = rnd(M, W, seed=94).to(DEV) * 0.01 if form in ("res1", "res12") else None
    res2 = (rnd(M, W, seed=95) * 0.01).bfloat16().to(DEV) if form == "res12" else None
    if form in ("res1", "res12"):
        res1[3] = 0
        if res2 is not None:
            res2[3] = 0
    y = torch.empty(M, W, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    K.layernorm_fwd(x, g, b, 1e-5, y, mean, rstd)

    def run(q8):
        dx = None if form == "xbf16" else torch.full((M, W), float("nan"), device=DEV)
        dxb = torch.full((M, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        dg, db = torch.zeros(W, device=DEV), torch.zeros(W, device=DEV)
        q = torch.full((M, W), 7, dtype=torch.uint8, device=DEV) if q8 else None
        rs = torch.full((M + 1,), float("nan"), device=DEV) if q8 else None
        K.layernorm_bwd(dy, x, mean, rstd, g, dx, dx_bf16=dxb, res1=res1, res2=res2, dgamma=dg, dbeta=db, q8=q, row_scale=rs)
        return dx, dxb, dg, db, q, rs

    dx0, dxb0, dg0, db0, _, _ = run(False)
    dx1, dxb1, dg1, db1, q, rs = run(True)
    assert torch.equal(dxb0, dxb1) and (dx0 is None or torch.equal(dx0, dx1))
    assert torch.allclose(dg0, dg1, rtol=1e-5, atol=1e-6) and torch.allclose(db0, db1, rtol=1e-5, atol=1e-6)
    q_ref, rs_ref = K.quantize_fp8_rows(dxb1)
    assert torch.equal(q, q_ref) and torch.equal(rs[:M], rs_ref)
    assert float(rs[3]) == 1.0 and int(q[3].max()) == 0 and torch.isnan(rs[M])


def test_gemm_tn_split_counts_agree(K):
    """the weight-gradient kernel under forced contraction-range counts (TVTS_TN_SPLITS in the call's opts): every count gives the fp32 product
    (different summation trees: equal within fp32 accumulation error), the column sums ride along unchanged."""
    M, Na, Nb = 40000, 768, 512
    p, q = bf(rnd(M, Na, seed=110)).to(DEV), bf(rnd(M, Nb, seed=111)).to(DEV)
    ref = p.float().t().double().cpu() @ q.float().double().cpu()
    outs = []
    for sp in (0, 1, 5, 8, 24):
        out = torch.zeros(Na, Nb, device=DEV)
        cs = torch.zeros(Na, device=DEV)
        K.gemm_tn(p, q, out, accumulate=False, colsum=cs, splits=sp)
        assert rel(out, ref) < 2e-6, (sp, rel(out, ref))
        assert rel(cs, p.float().sum(0).double().cpu()) < 1e-5, sp
        outs.append(out)
    assert rel(outs[1], outs[4]) < 2e-6


def test_transpose_batched(K):
    """the batched bf16 transpose behind the transposed weight shadows: 16-byte path (dimensions multiples of 8) and the element
    path (ragged shapes), one launch over a tile table."""
    import numpy as np
    shapes = [(768, 2304), (1280, 640), (64, 64), (200, 136), (300, 75), (8, 1000)]
    CH = 1024
    offs, toffs, so, to = [], [], 0, 0
    for r, c in shapes:
        offs.append(so); toffs.append(to)
        so += -(-r * c // CH) * CH; to += -(-r * c // CH) * CH
    src = torch.zeros(so, dtype=torch.bfloat16, device=DEV)
    dst = torch.full((to,), float("nan"), dtype=torch.bfloat16, device=DEV)
    tiles = []
    for (r, c), o, t in zip(shapes, offs, toffs):
        src[o:o + r * c] = rnd(r, c, seed=100 + r).bfloat16().to(DEV).reshape(-1)
        for tr in range(-(-r // 64)):
            for tc in range(-(-c // 64)):
                tiles.append((o, t, r, c, tr, tc))
    rec = np.zeros(len(tiles), dtype=[("s", "<i8"), ("d", "<i8"), ("R", "<i4"), ("C", "<i4"), ("tr", "<i4"), ("tc", "<i4")])
    for i, tt in enumerate(tiles):
        rec[i] = tt
    table = torch.from_numpy(rec.view(np.uint8).copy()).to(DEV)
    K.transpose_batched(src, dst, table, len(tiles))
    for (r, c), o, t in zip(shapes, offs, toffs):
        assert torch.equal(dst[t:t + r * c].view(c, r), src[o:o + r * c].view(r, c).t()), (r, c)


def test_fp8_multi_tensor_quantisation(K):
    """all fp8 weights in three launches: the same bytes and scales as tvts_amax + tvts_quant_fp8 tensor by tensor; the e4m3 copy of
    the bf16 transposed shadow under the master's scale."""
    shapes = [(256, 128), (1280, 3840), (5120, 1280), (12, 8), (768, 768)]
    ws = [(rnd(r, c, seed=70 + i) * (0.02 + 0.3 * i)).to(DEV) for i, (r, c) in enumerate(shapes)]
    ws[3].zero_()                                            # an all-zero weight: scale 1, zero bytes
    scal = torch.full((3 * len(ws),), float("nan"), device=DEV)
    ents, qs, qts = [], [], []
    for i, w in enumerate(ws):
        wt = w.t().contiguous().bfloat16() if i % 2 == 0 else None
        q = torch.full(w.shape, 7, dtype=torch.uint8, device=DEV)
        qt = torch.full((w.shape[1], w.shape[0]), 7, dtype=torch.uint8, device=DEV) if wt is not None else None
        ents.append((w, q, wt, qt, scal[3 * i:3 * i + 1], scal[3 * i + 1:3 * i + 2], scal[3 * i + 2:3 * i + 3] if wt is not None else None))
        qs.append(q); qts.append((wt, qt))
    table, n = K.quantize_fp8_multi_table(ents, DEV)
    for _ in range(2):                                       # a second call starts from stale amax values
        K.quantize_fp8_multi(table, n)
    for i, w in enumerate(ws):
        q1, s1 = K.quantize_fp8(w)
        assert torch.equal(qs[i], q1) and float(scal[3 * i + 1]) == float(s1) and float(scal[3 * i]) == float(w.abs().max())
        wt, qt = qts[i]
        if wt is not None:
            am = scal[3 * i:3 * i + 1].clone()
            q2, s2 = K.quantize_fp8(wt, amax=am, amax_given=True)
            assert torch.equal(qt, q2) and float(scal[3 * i + 2]) == float(s1)
            deq = qt.cpu().view(torch.float8_e4m3fn).float() * float(s1)
            assert rel(deq, w.t().float()) < 0.05
    assert float(scal[3 * 3 + 1]) == 1.0 and int(qs[3].max()) == 0


def test_fp8_tensor_scale_mode_of_the_quantising_kernels(K):
    """Per-tensor (delayed) scaling: with `tscale` the row quantiser and the LayerNorm forms write every row under that one scale
    (bit patterns of torch.float8_e4m3fn on x / tscale, saturating), `amax` receives the tensor's max |x|, and
    tvts_fp8_update_scales turns maxima into scales."""
    x = (rnd(1233, 1280, seed=57) * torch.linspace(0.01, 30.0, 1233)[:, None]).bfloat16()
    ts = torch.tensor([0.05], device=DEV)
    am = torch.zeros(1, device=DEV)
    q, rs = K.quantize_fp8_rows(x.to(DEV), tscale=ts, amax=am)
    assert rs is None and abs(float(am) - float(x.float().abs().max())) == 0.0
    want = _e4m3(x.float() * (1.0 / torch.tensor(0.05, dtype=torch.float32)))
    assert torch.equal(q.cpu().view(torch.uint8), want.view(torch.uint8))
    # LayerNorm forward, tensor mode: same bytes as quantising its bf16 output under the scale
    xx, g, b = rnd(300, 768, seed=58).to(DEV), (1 + 0.1 * rnd(768, seed=59)).to(DEV), (0.1 * rnd(768, seed=60)).to(DEV)
    y = torch.empty(300, 768, dtype=torch.bfloat16, device=DEV)
    q8 = torch.empty(300, 768, dtype=torch.uint8, device=DEV)
    ts2, am2 = torch.tensor([0.02], device=DEV), torch.zeros(1, device=DEV)
    K.layernorm_fwd(xx, g, b, 1e-5, y, q8=q8, tscale=ts2, amax=am2)
    assert float(am2) == float(y.float().abs().max())
    qq, _ = K.quantize_fp8_rows(y, tscale=ts2)
    assert torch.equal(q8, qq)
    amax, scale = torch.tensor([4.48, 0.0, 896.0], device=DEV), torch.tensor([1.0, 7.0, 1.0], device=DEV)
    K.fp8_update_scales(amax, scale)
    assert torch.allclose(scale.cpu(), torch.tensor([0.01, 7.0, 2.0])) and float(amax.abs().max()) == 0.0


@pytest.mark.parametrize("rows,cols", [(300, 256), (1233, 1280), (77, 5120), (9, 5128), (5, 8)])
def test_fp8_row_quantisation(K, rows, cols):
    """per-row (token) e4m3 quantisation in one pass: row_scale = amax(row) / 448, bit patterns of torch.float8_e4m3fn on
    x * (1 / row_scale); an all-zero row gets scale 1 and zero bytes."""
    x = (rnd(rows, cols, seed=56) * torch.linspace(0.01, 30.0, rows)[:, None]).bfloat16()
    x[0] = 0
    x[min(3, rows - 1), cols - 1] = -57.0
    q, rs = K.quantize_fp8_rows(x.to(DEV))
    s = x.float().abs().amax(dim=1) / 448.0
    s[0] = 1.0
    assert torch.allclose(rs.cpu(), s, rtol=1e-6, atol=0)
    want = _e4m3(x.float() * (1.0 / rs.cpu())[:, None])
    got = q.cpu().view(torch.float8_e4m3fn)
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))
    assert int(q[0].max()) == 0
    # strided input / output views (columns of wider matrices), preallocated outputs
    wide = torch.zeros(rows, cols + 64, dtype=torch.bfloat16, device=DEV)
    wide[:, 32:32 + cols] = x.to(DEV)
    if cols % 8 == 0 and 32 % 8 == 0:
        qw = torch.full((rows, cols + 16), 7, dtype=torch.uint8, device=DEV)
        q2, rs2 = K.quantize_fp8_rows(wide[:, 32:32 + cols], q=qw[:, 8:8 + cols], row_scale=torch.empty(rows + 3, device=DEV))
        assert torch.equal(q2, q) and torch.equal(rs2[:rows], rs)
        assert int(qw[:, :8].min()) == 7 and int(qw[:, 8 + cols:].min()) == 7


@pytest.mark.parametrize("M,N,K_", [(256, 256, 128), (777, 1280, 1280), (9420, 3072, 768)])
def test_gemm_nt_fp8_row_scales(K, M, N, K_):
    """out = row_scale[m] * sb * (A8 B8^T) + bias with the per-token scales of quantize_fp8_rows."""
    a = rnd(M, K_, seed=57) * torch.logspace(-5, 1.5, M)[:, None]     # rows of very different magnitude (the small ones underflow a tensor-wide scale)
    b = rnd(N, K_, seed=58) * K_ ** -0.5
    a8, rs = K.quantize_fp8_rows(a.bfloat16().to(DEV))
    b8, sb = K.quantize_fp8(b.to(DEV))
    bias = rnd(N, seed=59).to(DEV)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    K.gemm_nt_fp8(a8, rs, b8, sb, out, bias=bias)
    ad, bd = a8.cpu().view(torch.float8_e4m3fn).float(), b8.cpu().view(torch.float8_e4m3fn).float()
    ref = (ad.double() @ bd.double().t()) * rs.cpu().double()[:, None] * float(sb) + bias.cpu().double()
    err = ((out.cpu().double() - ref).abs() / (ref.abs().amax(dim=1, keepdim=True) + 1e-30)).max()
    assert float(err) < 5e-5, float(err)
    # per-token scales keep the small rows accurate: row-relative error against the unquantised product
    exact = a.bfloat16().double() @ b.double().t()
    rowrel = ((out.cpu().double() - bias.cpu().double() - exact).norm(dim=1) / exact.norm(dim=1))
    assert float(rowrel.max()) < 0.08, float(rowrel.max())
    # one per-tensor scale loses them (what the per-row scales are for)
    a8t, sat = K.quantize_fp8(a.bfloat16().to(DEV))
    out_t = torch.empty_like(out)
    K.gemm_nt_fp8(a8t, sat, b8, sb, out_t)
    rowrel_t = ((out_t.cpu().double() - exact).norm(dim=1) / exact.norm(dim=1))
    assert float(rowrel_t.max()) > 3 * float(rowrel.max())
    act_out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt_fp8(a8, rs, b8, sb, act_out, bias=bias, act="gelu", preact=pre)
    assert rel(pre.float(), ref) < 4e-3
    assert rel(act_out.float(), torch.nn.functional.gelu(ref.float()).double()) < 6e-3


@pytest.mark.parametrize("W,xdt", [(768, torch.float32), (1280, torch.float32), (1280, torch.bfloat16), (320, torch.float32), (1024, torch.bfloat16)])
def test_layernorm_fwd_fp8_output(K, W, xdt):
    """LayerNorm forward with the fused e4m3 copy: y / mean / rstd identical to the plain kernel, the fp8 bytes and per-row
    scales identical to quantize_fp8_rows(y)."""
    M = 517
    x = (rnd(M, W, seed=60) * 2 + 0.3).to(xdt).to(DEV)
    g, b = (1 + 0.1 * rnd(W, seed=61)).to(DEV), (0.1 * rnd(W, seed=62)).to(DEV)
    y0, y1 = (torch.empty(M, W, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    m0, r0, m1, r1 = (torch.empty(M, device=DEV) for _ in range(4))
    K.layernorm_fwd(x, g, b, 1e-5, y0, m0, r0)
    q = torch.full((M, W), 3, dtype=torch.uint8, device=DEV)
    rs = torch.empty(M, device=DEV)
    K.layernorm_fwd(x, g, b, 1e-5, y1, m1, r1, q8=q, row_scale=rs)
    if xdt == torch.float32:
        assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)
    else:  # bf16 rows: the plain forward is the 8-columns-per-lane kernel (round 5), another summation order than the fused-copy form
        assert torch.allclose(m0, m1, rtol=0, atol=1e-6) and torch.allclose(r0, r1, rtol=1e-6, atol=0)
        assert float((y0.float() - y1.float()).abs().max()) <= 2.0 ** -7 * float(y0.float().abs().max())  # a last bf16 bit here and there
    q_ref, rs_ref = K.quantize_fp8_rows(y1)
    assert torch.equal(q, q_ref) and torch.equal(rs, rs_ref)
    if xdt == torch.bfloat16:  # the e4m3 bytes as the ONLY output (arch["fp8_q8_only"]): the same bytes, per-row and per-tensor scales
        assert torch.equal(y0, y1)  # (both are the 8-column kernel now)
        q3, rs3, m3, r3 = torch.full_like(q, 5), torch.empty_like(rs), torch.empty_like(m1), torch.empty_like(r1)
        K.layernorm_fwd(x, g, b, 1e-5, None, m3, r3, q8=q3, row_scale=rs3)
        assert torch.equal(q3, q) and torch.equal(rs3, rs) and torch.equal(m3, m1) and torch.equal(r3, r1)
        ts = torch.tensor([float(y1.float().abs().max()) / 448.0 * 0.9], device=DEV)  # (0.9: some values clamp at +-448)
        q4, q5 = torch.full_like(q, 5), torch.full_like(q, 6)
        am4, am5 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
        K.layernorm_fwd(x, g, b, 1e-5, torch.empty_like(y1), q8=q4, tscale=ts, amax=am4)
        K.layernorm_fwd(x, g, b, 1e-5, None, q8=q5, tscale=ts, amax=am5)
        assert torch.equal(q4, q5) and torch.equal(am4, am5) and float(am4) == float(y1.float().abs().max())
        want = torch.clamp(y1.float() * (1.0 / ts), -448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(q5, want)
    else:
        with pytest.raises(K.HipError):  # fp32 rows have no e4m3-only form
            K.layernorm_fwd(x, g, b, 1e-5, None, m1, r1, q8=q, row_scale=rs)
    # gathered rows (the pooled tail's CLS rows)
    rows = torch.tensor([5, 0, 333, 516], dtype=torch.int32, device=DEV)
    y2 = torch.empty(4, W, dtype=torch.bfloat16, device=DEV)
    q2, rs2 = torch.empty(4, W, dtype=torch.uint8, device=DEV), torch.empty(4, device=DEV)
    K.layernorm_fwd(x, g, b, 1e-5, y2, torch.empty(4, device=DEV), torch.empty(4, device=DEV), rows=rows, q8=q2, row_scale=rs2)
    assert torch.equal(y2, y1[rows.long()]) and torch.equal(q2, q_ref[rows.long()]) and torch.equal(rs2, rs_ref[rows.long()])


@pytest.mark.parametrize("M,Na,Nb", [(5000, 768, 3072), (4097, 256, 128), (12345, 1280, 640)])
def test_gemm_tn_dma_paths_agree(K, M, Na, Nb):
    """The weight-gradient kernel issues its LDS-DMA from inline asm ahead of the fragment reads (32-bit offsets from a
    uniform base); operands past 4 GiB take the builtin path.  Both paths and both tile walks must give identical bits."""
    p, q = rnd(M, Na, seed=70).bfloat16().to(DEV), rnd(M, Nb, seed=71).bfloat16().to(DEV)
    outs = []
    for early, afast in ((None, None), (False, None), (None, False), (None, True)):
        out = torch.full((Na, Nb), float("nan"), device=DEV)
        cs = torch.zeros(Na, device=DEV)
        K.gemm_tn(p, q, out, accumulate=False, colsum=cs, early_dma=early, a_fast=afast)
        outs.append((out, cs))
    ref = p.float().t().double() @ q.float().double()
    assert rel(outs[0][0], ref.cpu()) < 2e-5
    for o, c in outs[1:]:
        assert torch.equal(o, outs[0][0])
        assert torch.equal(c, outs[0][1])   # column sums: ordered per-range partials, whatever the DMA path / tile walk
    assert torch.allclose(outs[0][1].cpu(), p.float().sum(0).cpu(), rtol=1e-3, atol=2e-2)


def test_two_threads_call_different_shapes_and_options_concurrently(K):
    """backward kernels are launched from PyTorch's autograd worker thread (and communication hooks fire there)
    while the main thread runs forward / optimizer.  Two host threads, each on its own stream, hammer the GEMM, LayerNorm and
    attention entry points with DIFFERENT shapes and different per-call dispatch options (forced tiles, split counts, fused /
    split attention) -- ctypes drops the GIL inside the calls, so the dispatch code really runs concurrently.  Every result must
    be bit-identical to the same call made alone: the library keeps no state one caller could change under the other."""
    import threading

    def work(seed, M, N, Kd, tile, tn_tile, splits, fused, reps, streams, out):
        try:
            torch.cuda.set_device(0)
            s = streams[seed]
            with torch.cuda.stream(s):
                g = torch.Generator(device=DEV).manual_seed(seed)
                a = torch.randn(M, Kd, generator=g, device=DEV).bfloat16()
                b = (torch.randn(N, Kd, generator=g, device=DEV) * Kd ** -0.5).bfloat16()
                bias = torch.randn(N, generator=g, device=DEV)
                B_, h, T, n = 2, 2, 3, 21 + seed
                S, W = 1 + T * n, h * 64
                qkv = torch.randn(B_ * S, 3 * W, generator=g, device=DEV).bfloat16()
                dO = torch.randn(B_ * S, W, generator=g, device=DEV).bfloat16()
                res = []
                for _ in range(reps):
                    o = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
                    K.gemm_nt(a, b, o, bias=bias, act="quick_gelu", preact=torch.empty_like(o), tile=tile)
                    w = torch.zeros(N, Kd, device=DEV)
                    cs = torch.zeros(N, device=DEV)
                    K.gemm_tn(o, a, w, accumulate=False, colsum=cs, tile=tn_tile, splits=splits, workspace=False)
                    att = torch.empty(B_ * S, W, dtype=torch.bfloat16, device=DEV)
                    lse = torch.empty(B_ * S, h, device=DEV)
                    ws = torch.empty(B_ * h * max(T, -(-n // 28)) * 66, device=DEV)
                    K.attn_fwd_divided("space", qkv, att, lse, ws, B=B_, heads=h, S=S, T=T, n=n, fused=fused)
                    dqkv = torch.empty_like(qkv)
                    delta = torch.empty(B_ * S, h, device=DEV)
                    acc = torch.empty(B_, h, T, 3, 64, device=DEV)
                    K.attn_bwd("space", qkv, dO, att, lse, delta, dqkv, B=B_, heads=h, S=S, T=T, n=n, cls_acc=acc, fused=fused)
                    res.append((o, w, cs, att, dqkv))
                s.synchronize()
            out[seed] = res
        except Exception as e:  # surfaced by the main thread
            out[seed] = e

    # the TN calls run with splits = 1 and no workspace: the shared module-level workspace tensor of hip.gemm_tn is the caller's
    # buffer (one per stream in a multi-stream caller), not library state
    cfg = {0: (3000, 768, 256, 128, 128, 1, True), 1: (5000, 512, 192, 256, 256, 1, False)}
    streams = {0: torch.cuda.Stream(), 1: torch.cuda.Stream()}
    alone = {}
    for sd, c in cfg.items():
        work(sd, *c, 1, streams, alone)
        assert not isinstance(alone[sd], Exception), alone[sd]
    both = {}
    th = [threading.Thread(target=work, args=(sd, *c, 20, streams, both)) for sd, c in cfg.items()]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    for sd in cfg:
        assert not isinstance(both[sd], Exception), both[sd]
        for r in both[sd]:
            for x, y in zip(r, alone[sd][0]):
                assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y)
