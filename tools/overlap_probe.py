"""Can a side-stream kernel (the RCCL kernels of the data-parallel step) run under the backward GEMM chain on ONE GPU?

The persistent 256x256 NT kernel occupies a CU completely (160 KiB of LDS, 2 x 256 registers per SIMD lane): a kernel of another
stream has no CU to land on until a GEMM block retires -- a high-priority stream does not pre-empt resident workgroups.  This probe
runs the step's dgrad GEMM chain (M = pairs x 785 token rows) on the compute stream and, forked by an event behind the third
GEMM, a copy kernel of RCCL-like size (64 MiB, grid of 256-thread blocks without LDS) on a high-priority non-blocking side stream
-- the kind of stream libtvts_comm.so owns -- and reports
  * the copy's completion latency measured from the fork event (and its duration alone on an idle chip),
  * the GEMM chain's duration with / without the concurrent copy,
for persistent grids of 256 / 248 / 240 / 224 CUs (TVTS_GEMM_CUS in the call's opts).  What dist.py reserves for world > 1 comes
from this table (profiles/r03_overlap_probe.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvts_amd import hip as K  # noqa: E402

dev = "cuda:0"
PAIRS = int(os.environ.get("PAIRS", "192"))
M = PAIRS * 785
shapes = [(768, 3072), (3072, 768), (768, 768), (768, 2304), (768, 768), (768, 2304)] * 4   # (N, K) of fc2 / fc1 / proj / qkv dgrads
g = torch.Generator(device=dev).manual_seed(0)
A_ = {k: torch.randn(M, k, generator=g, device=dev).bfloat16() for k in (768, 2304, 3072)}
W_ = {(n, k): (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16() for (n, k) in set(shapes)}
O_ = {n: torch.empty(M, n, dtype=torch.bfloat16, device=dev) for n in (768, 3072)}
src = torch.randn(16 * 1024 * 1024, device=dev)
dst = torch.empty_like(src)
side = torch.cuda.Stream(priority=-1)


def chain(cus):
    for (n, k) in shapes:
        K.gemm_nt(A_[k], W_[(n, k)], O_[n], cus=cus)


def run(cus, with_copy, fork_after=3):
    main = torch.cuda.current_stream()
    e0, e1, ef, s0, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(5))
    torch.cuda.synchronize()
    e0.record()
    for i, (n, k) in enumerate(shapes):
        K.gemm_nt(A_[k], W_[(n, k)], O_[n], cus=cus)
        if i + 1 == fork_after:
            ef.record()
            if with_copy:
                side.wait_event(ef)
                with torch.cuda.stream(side):
                    s0.record()
                    dst.copy_(src)
                    s1.record()
    if with_copy:
        main.wait_event(s1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (ef.elapsed_time(s1), s0.elapsed_time(s1)) if with_copy else None


# the copy alone
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    with torch.cuda.stream(side):
        a.record(); dst.copy_(src); b.record()
    torch.cuda.synchronize()
print(f"copy of {src.numel() * 4 >> 20} MiB alone on the idle chip: {a.elapsed_time(b) * 1e3:.0f} us")
print(f"GEMM chain: {len(shapes)} dgrad launches at M = {M}")
for cus in (256, 248, 240, 224):
    for _ in range(2):
        run(cus, False); run(cus, True)
    base = sorted(run(cus, False)[0] for _ in range(5))[2]
    rs = [run(cus, True) for _ in range(7)]
    tot = sorted(r[0] for r in rs)[3]
    lat = sorted(r[1][0] for r in rs)
    dur = sorted(r[1][1] for r in rs)
    print(f"grid {cus:3d} CUs: chain alone {base:7.3f} ms | with the side-stream copy {tot:7.3f} ms | copy done {lat[3] * 1e3:7.0f} us after the fork "
          f"(min {lat[0] * 1e3:.0f}, max {lat[-1] * 1e3:.0f}); first-block-to-end {dur[3] * 1e3:7.0f} us", flush=True)
