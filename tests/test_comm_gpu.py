"""The C-ABI exchange steps (include/tvts_comm.h, libtvts_comm.so) on the one GPU of the test box: RCCL with a
world of one rank -- every entry point, the side stream's fork / join against the compute stream, and a whole
training step routed through it (TVTS_COMM=native) against the torch.distributed transport.  (RCCL refuses two ranks on
one device, so world > 1 of this transport runs only where the driver has more GPUs; the exchange semantics at world 2
are covered on gloo by tests/test_dist_cpu.py and tests/test_dist_gpu.py.)"""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tvts_oracle as O  # noqa: E402  (synthetic batches / parameters only)

DEV = "cuda:0"


@pytest.fixture(scope="module")
def comm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tvts_amd import dist as D
    return D.NativeComm.get()


def test_native_comm_world_of_one(comm):
    assert (comm.W, comm.rank) == (1, 0)
    g = torch.Generator(device=DEV).manual_seed(0)
    v, t = torch.randn(48, 512, generator=g, device=DEV), torch.randn(48, 512, generator=g, device=DEV)
    va, ta = torch.full_like(v, float("nan")), torch.full_like(t, float("nan"))
    comm.allgather_embeds(v, t, va, ta)
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(va, v) and torch.equal(ta, t)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(1 << 20, generator=g, device=DEV).to(dt)
        ref = x.clone()
        comm.allreduce(x)
        comm.wait()
        torch.cuda.synchronize()
        assert torch.equal(x, ref)


def test_known_answer_self_test(comm):
    """the check `dist.transport()` runs on every rank before it settles on the native transport (a wrong or failing rank sends
    all of them to torch.distributed instead)"""
    comm.self_test()


def test_side_stream_orders_against_the_compute_stream(comm):
    """producer kernel -> collective -> consumer kernel with no host synchronisation in between, many times over: the
    collective must see the producer's output (fork) and the consumer the collective's (join)."""
    n = 1 << 22
    x = torch.zeros(n, device=DEV)
    acc = torch.zeros(n, device=DEV)
    for i in range(50):
        x.add_(1.0)            # producer on the compute stream
        comm.allreduce(x)      # side stream, world 1: identity
        comm.wait()
        acc.add_(x)            # consumer on the compute stream
    torch.cuda.synchronize()
    assert float(x[0]) == 50.0 and float(acc.min()) == float(acc.max()) == 50 * 51 / 2


def _step(native: bool, payload: str):
    from tvts_amd import arch as A
    from tvts_amd import dist as D
    from tvts_amd.model._common import TVTSv2Base
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    a = A.small_arch()
    oarch = O.tiny_arch(**a)
    m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
    m.load_state_dict(O.synth_params(oarch, seed=1), strict=True)
    groups = [[], [], [], []]
    for name, p in m.named_parameters():
        gi = A.param_group_of(name, a)
        if gi < 0:
            p.requires_grad = False
        else:
            groups[gi].append(p)
    opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0] * 30, weight_decay=A.GROUP_HPARAMS[i][1])
                        for i in range(4)], m.store, model=m)
    run = StepRunner(m, opt)
    run.sync = D.GradSync(m.store.grad, payload=payload, native=native)
    run.gather = D.EmbedGather(native=native)
    m.engine.grad_ready = run.sync.reduce_range if native else None
    batch = O.synth_batch(oarch, B=4, T=2, seed=2, caption_len=9)
    losses = []
    for _ in range(3):
        out = run.step(batch)
        losses.append(float(out["loss1"]) + float(out["loss2"]))
    torch.cuda.synchronize()
    return losses, m.store.flat.clone(), run.sync.bytes_sent, m


def test_training_step_through_the_native_transport(comm):
    l0, p0, sent0, _ = _step(False, "fp32")
    l1, p1, sent1, m = _step(True, "fp32")
    assert sent0 == 0 and sent1 > 0
    trainable = sum(-(-p.numel() // 1024) * 1024 for p in m.parameters() if p.requires_grad)
    assert sent1 == 4 * trainable  # every trainable range exactly once per step, frozen ranges never
    assert l1 == pytest.approx(l0, rel=1e-6, abs=1e-6)
    assert float((p1 - p0).abs().max()) < 2e-6
    # bf16 payload: gradients rounded to bf16 on the wire, half the bytes
    l2, p2, sent2, _ = _step(True, "bf16")
    assert sent2 * 2 == sent1
    assert l2[0] == pytest.approx(l0[0], rel=1e-6, abs=1e-6) and l2[-1] == pytest.approx(l0[-1], rel=5e-3, abs=5e-3)


def test_captured_step_through_the_native_transport(comm):
    """The step with its exchange steps on the library's side stream, captured into ONE hipGraph (the fork / join events pull
    the side stream into the capture; RCCL launches are capturable) and replayed: the same losses and parameters as eager
    launches of the same step.  What bench.py does under TVTS_BENCH_GRAPH_DDP=1 for world > 1."""
    from tvts_amd import arch as A
    from tvts_amd import dist as D
    from tvts_amd.model._common import TVTSv2Base
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner
    a = A.small_arch()
    oarch = O.tiny_arch(**a)
    batch = O.synth_batch(oarch, B=4, T=2, seed=2, caption_len=9)

    def make():
        m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=0, world_size=1), arch=a)
        m.load_state_dict(O.synth_params(oarch, seed=1), strict=True)
        groups = [[], [], [], []]
        for name, p in m.named_parameters():
            gi = A.param_group_of(name, a)
            if gi < 0:
                p.requires_grad = False
            else:
                groups[gi].append(p)
        opt = FusedHFAdamW([dict(params=groups[i], lr=A.GROUP_HPARAMS[i][0] * 30, weight_decay=A.GROUP_HPARAMS[i][1])
                            for i in range(4)], m.store, model=m)
        run = StepRunner(m, opt)
        run.sync = D.GradSync(m.store.grad, native=True)
        run.gather = D.EmbedGather(native=True)
        m.engine.grad_ready = run.sync.reduce_range
        pb = m.engine.prepare_batch(batch)
        lab = batch["label"].reshape(-1).to(torch.int32).to(DEV)
        m._fresh_shadows(); m._sync_requires_grad()
        return m, opt, run, pb, lab

    m, opt, run, pb, lab = make()
    eager = []
    for _ in range(3):
        out = run.run(pb, lab, device_step=True)
        torch.cuda.synchronize()
        eager.append(float(out["loss1"]) + float(out["loss2"]))
    p_eager = m.store.flat.clone()

    m, opt, run, pb, lab = make()
    snap = {k: t.clone() for k, t in (("flat", m.store.flat), ("m", m.store.m), ("v", m.store.v))}
    run.run(pb, lab, device_step=True)  # warm-up: workspaces, communicator, hyper-parameter table
    torch.cuda.synchronize()
    m.store.flat.copy_(snap["flat"]); m.store.m.copy_(snap["m"]); m.store.v.copy_(snap["v"])
    opt.step_dev.zero_(); opt.global_step = 0
    m.store.refresh_shadows()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run.run(pb, lab, device_step=True)
    torch.cuda.synchronize()
    assert torch.equal(m.store.flat, snap["flat"])  # capture does not execute
    replayed = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        replayed.append(float(out["loss1"]) + float(out["loss2"]))
    assert run.sync.bytes_sent > 0
    assert replayed == eager, (replayed, eager)
    assert float((m.store.flat - p_eager).abs().max()) < 1e-7


def test_bring_up_has_a_deadline_and_a_dead_peer_does_not_hang_the_survivor(comm):
    """tvts_comm_create_deadline: the rendezvous of a 2-rank communicator whose second rank never arrives (a peer that died after
    the id broadcast) returns TVTS_COMM_ETIMEDOUT after the deadline instead of blocking inside ncclCommInitRank for ever -- the
    survivor is free to agree on the torch.distributed transport (dist.transport()).  tvts_comm_idle polls the side stream
    without blocking; tvts_comm_abort tears a communicator down without waiting for its collectives."""
    import ctypes
    import time
    lib = comm.lib
    raw = (ctypes.c_ubyte * 128)()
    assert lib.tvts_comm_unique_id(ctypes.cast(raw, ctypes.c_void_p)) == 0
    h = ctypes.c_void_p()
    t0 = time.time()
    rc = lib.tvts_comm_create_deadline(ctypes.cast(raw, ctypes.c_void_p), 0, 2, 1500, ctypes.byref(h))
    dt = time.time() - t0
    assert rc == -110 and not h.value, (rc, h.value)
    assert 1.0 < dt < 20.0, dt
    # the device and the existing communicator are unharmed by the parked helper thread
    x = torch.ones(1 << 16, device=DEV)
    comm.allreduce(x)
    assert comm.drain(30.0)
    comm.wait()
    torch.cuda.synchronize()
    assert float(x.sum()) == float(1 << 16)
    assert lib.tvts_comm_idle(comm.h) == 1
    # a second communicator of one rank, with a deadline that is met, then aborted instead of destroyed
    raw2 = (ctypes.c_ubyte * 128)()
    assert lib.tvts_comm_unique_id(ctypes.cast(raw2, ctypes.c_void_p)) == 0
    h2 = ctypes.c_void_p()
    assert lib.tvts_comm_create_deadline(ctypes.cast(raw2, ctypes.c_void_p), 0, 1, 60000, ctypes.byref(h2)) == 0 and h2.value
    assert lib.tvts_comm_idle(h2) == 1
    assert lib.tvts_comm_abort(h2) == 0
