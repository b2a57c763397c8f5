#!/usr/bin/env python3
"""Two data-parallel ranks of the HIP step engine on ONE GPU (gloo backend moving CUDA tensors), checked against the
oracle's multi-rank step: all-gather of embeddings, local-row backward, range-by-range gradient all-reduce.
usage: python tools/dist_smoke.py [h14]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, h14, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from oracle import tvts_oracle as O
    from tvts_amd import arch as A
    from tvts_amd.model._common import TVTSv2Base
    from tvts_amd.step import StepRunner
    a = A.small_arch_h() if h14 else A.small_arch()
    oarch = O.tiny_arch(**a)
    P = O.synth_params(oarch, seed=3)
    m = TVTSv2Base(types.SimpleNamespace(local_rank=0, rank=rank, world_size=world), arch=a)
    m.load_state_dict(P, strict=True)

    class NoOpt:  # the step without a parameter update: gradients stay in the flat buffer
        def step(self):
            pass
    runner = StepRunner(m, NoOpt())
    batches = [O.synth_batch(oarch, B=3, T=3, seed=50 + r, caption_len=9) for r in range(world)]
    out = runner.step(batches[rank])
    torch.cuda.synchronize()
    # the data-parallel invariant: after the all-reduce every rank holds the SAME averaged gradient, bit for bit (the replicas
    # would drift apart otherwise), and the same global contrastive loss
    chk = torch.stack([m.store.grad.view(torch.int32).to(torch.int64).sum(), out["loss1"].reshape(()).view(torch.int32).to(torch.int64)])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    same = all(torch.equal(both[0], b) for b in both[1:])
    if rank == 0:
        leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        total, loss1, l2 = O.multi_rank_step(leaves, batches, oarch)
        total.backward()
        ref = {k: v.grad for k, v in leaves.items() if v.grad is not None}
        grads = {k: m.store.g(k).detach().cpu() for k in P}
        l1e, l2e, r1, r2 = float(out["loss1"]), float(out["loss2"]), float(loss1), float(l2[0])
        print("engine losses (rank 0)", l1e, l2e, "oracle", r1, 2 * 0 + r2, flush=True)
        tot_r = sum(float(v.norm()) ** 2 for v in ref.values()) ** .5
        tot = sum(float(grads[k].norm()) ** 2 for k in ref) ** .5
        worst = min((float(torch.nn.functional.cosine_similarity(grads[k].flatten().double(), v.flatten().double(), dim=0)), k)
                    for k, v in ref.items() if float(v.norm()) > 1e-3 * tot_r)
        print("grad norm", tot, tot_r, "worst cos", worst, flush=True)
        print("ranks hold the same gradient bits and loss:", same, flush=True)
        ok = same and abs(l1e - r1) < 1e-2 and abs(tot - tot_r) < 0.02 * tot_r and worst[0] > 0.98
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    h14 = len(sys.argv) > 1 and sys.argv[1] == "h14"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29511, h14, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join()
    assert ok and all(p.exitcode == 0 for p in procs)
    print("DIST_SMOKE_OK")
