"""World-size-2 gloo test of the data-parallel exchange steps (tvts_amd/dist.py) on CPU tensors, with the
oracle as the compute, against the reference's own 2-rank run (tests/golden/ddp2_tiny.npz)."""
import os

import numpy as np
import torch
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ddp2_tiny.npz")


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import tvts_oracle as O
    from tvts_amd import dist as D
    f = np.load(GOLD)
    arch = O.tiny_arch()
    P = {k: v.clone().requires_grad_(True) for k, v in O.synth_params(arch, seed=int(f["seed"])).items()}
    batch = O.synth_batch(arch, B=int(f["B"]), T=int(f["T"]), seed=int(f[f"batch_seed{rank}"]), caption_len=int(f["caption_len"]))
    te, ve, pred = O.model_forward(P, batch, arch)
    B = ve.shape[0]
    assert D.world() == (world, rank)
    v_all, t_all = D.allgather_embeds(ve.detach(), te.detach())
    v_all.requires_grad_(True); t_all.requires_grad_(True)
    loss1 = O.norm_softmax_loss(O.sim_matrix(v_all, t_all))
    loss1.backward()
    # AllGather_multi.backward: local rows only, no reduction
    loss2 = O.sorting_ce(pred, batch["label"])
    torch.autograd.backward([ve, te, loss2], [D.local_rows(v_all.grad, B), D.local_rows(t_all.grad, B), torch.ones(())])
    # flat gradient buffer + asynchronous range reductions, 1/W applied afterwards
    names = [k for k in P if P[k].grad is not None]
    sizes = [P[k].numel() for k in names]
    flat = torch.cat([P[k].grad.reshape(-1) for k in names])
    flat16 = flat.clone()
    sync = D.GradSync(flat, bucket_bytes=64 << 10)
    cut = flat.numel() // 3
    sync.reduce_range(cut, flat.numel())
    sync.reduce_range(0, cut)
    flat.mul_(sync.finish())
    # the same exchange with the bf16 payload: half the bytes, the fp32 result to bf16 rounding
    sync16 = D.GradSync(flat16, bucket_bytes=64 << 10, payload="bf16")
    sync16.reduce_range(cut, flat16.numel())
    sync16.reduce_range(0, cut)
    flat16.mul_(sync16.finish())
    assert sync16.bytes_sent * 2 == sync.bytes_sent == flat.numel() * 4
    err16 = float((flat16 - flat).norm() / flat.norm())
    assert 0 < err16 < 6e-3, err16
    # reduce_range hands back a wait() for ITS range (the optimizer's range-wise update calls it on its own stream in front of that
    # range's AdamW launch): waiting early completes and removes exactly those reductions, finish() takes the rest, same sums
    flat_w = torch.cat([P[k].grad.reshape(-1) for k in names])
    for payload in ("fp32", "bf16"):
        fw = flat_w.clone()
        sw = D.GradSync(fw, bucket_bytes=64 << 10, payload=payload)
        w_hi = sw.reduce_range(cut, fw.numel())
        n_hi = len(sw.handles)
        w_lo = sw.reduce_range(0, cut)
        assert callable(w_hi) and callable(w_lo) and len(sw.handles) > n_hi > 0
        w_hi()
        assert len(sw.handles) == len(sw.handles) and all(s < cut for _, s, _ in sw.handles)  # only the low range is left
        w_hi()  # idempotent
        fw.mul_(sw.finish())
        assert not sw.handles
        assert torch.equal(fw, flat if payload == "fp32" else flat16), payload
    # the embedding exchange through the preallocated EmbedGather gives what allgather_embeds gives
    eg = D.EmbedGather()
    eg.start(te.detach(), ve.detach())
    v2, t2 = eg.result()
    assert torch.equal(v2, v_all.detach()) and torch.equal(t2, t_all.detach())
    out, o = {}, 0
    for k, n in zip(names, sizes):
        out[k] = flat[o:o + n].view_as(P[k]).clone().numpy(); o += n
    q.put((rank, float(loss1), float(loss2), float(flat.norm()), {k: out[k] for k in out if "g_" + k in f.files}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_reference_ddp():
    f = np.load(GOLD)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    [p.join() for p in procs]
    for r in range(2):
        assert abs(res[r][1] - float(f["loss1"][r])) < 1e-5   # every rank sees the same global InfoNCE
        assert abs(res[r][2] - float(f["loss2"][r])) < 1e-5   # the sorting loss is local
        assert abs(res[r][3] - float(f["grad_norm"])) < 1e-4 * float(f["grad_norm"])
        for k, g in res[r][4].items():
            ref = f["g_" + k]
            assert np.linalg.norm(g - ref) < 1e-4 * np.linalg.norm(ref), k


def test_cu_reservation_rule():
    """dist.auto_cu_reservation: what the persistent GEMM grids leave to the RCCL kernels, from the per-GPU token rows and the
    world size (DESIGN.md section 6 records the model and the prediction the first multi-GPU run has to confirm)."""
    from tvts_amd.dist import auto_cu_reservation as rule
    assert rule(150720, 1) == 0 and rule(9420, 1) == 0      # one GPU: nothing to overlap with
    assert rule(9420, 8) == 0                                 # 12 pairs: 128 x 128 kernels, RCCL's workgroups co-reside
    assert rule(18840, 8) == 8 and rule(37680, 8) == 8        # 24 / 48 pairs: every CU held by a persistent block, traffic 10-25 % of the backward
    assert rule(150720, 8) == 0                               # 192 pairs: the all-reduce is a few per cent of the backward
    assert rule(18840, 2) in (0, 8)                           # two ranks move half of it: either side of the threshold is fine
    assert rule(37680, 8, grad_bytes=312 << 20) in (0, 8)
