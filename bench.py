#!/usr/bin/env python3
"""TVTSv2 pretrain-step benchmark (BASELINE.json metric): video-text pairs/sec, ViT-B/16, 8 frames, mask 0.5,
32-token captions, NT=4 captions per video, bf16 MFMA compute, synthetic data, random-init weights.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = zero_grad, forward (text tower, space-time ViT, sort head), all-gather of embeddings, InfoNCE + 2*CE,
backward, gradient all-reduce, fused HF-AdamW -- on a batch that is already resident in HBM.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (measured 2495)
PEAK_HBM_BPS = 8.0e12        # HBM3E peak, MI355X_MICROARCH.md
ACHIEVABLE_HBM_BPS = 6.3e12  # what a streaming copy reaches on this part (same guide)
# world 1: below this many (ViT token rows per GPU) x width^2 the eager step is bound by the host's launch rate (~800 launches of ViT-B/16 take
# 11.5 ms however little they compute) and the captured graph is replayed by default: B/16 at 2 / 4 pairs replayed 212 / 374 pairs/s against
# 163 - 170 / 331 - 347 eager; at 6 pairs (2.8e9) and for H/14 at its 2 pairs (4.0e9) eager wins by 2 / 7.5 % (profiles/r06_bench_launch_bound_eager_vs_graph.txt)
GRAPH_AUTO_WORK = 2.5e9
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 peak (MX-scaled K = 128 MFMA; the non-scaled 16x16x32 fp8 form issues at the bf16 rate)


def step_flops_per_pair(a, T, caption_len=32, NT=4):
    """SURVEY.md 8d algorithmic matmul FLOPs per pair (fwd, bwd).  NT = 1 (WebVid-style batch): no sorting head."""
    p, W, E, Wt = a["patch"], a["width"], a["embed"], a["text_width"]
    n = int((a["image"] // p) ** 2 * (1 - a["mask_ratio"]))
    S, layers, Lt, L = 1 + T * n, a["layers"], a["text_layers"], caption_len
    So = S + NT
    patch = 2 * T * n * 3 * p * p * W
    text_gemm = NT * Lt * L * 24 * Wt * Wt
    sort = (2 * So * 24 * E * E + 8 * So * So * E + 32 * E) if NT > 1 else 0
    fwd = (patch + layers * S * 32 * W * W + layers * (4 * n * T * (T + 1) * W + 4 * T * n * (n + 1) * W + 8 * S * W)
           + 2 * S * W * E + text_gemm + NT * Lt * 4 * L * L * Wt + NT * 2 * Wt * E + sort)
    bwd = 2 * fwd - patch - (a["text_tune_from"] / Lt) * text_gemm
    return fwd, bwd


def step_flops_per_pair_v1(a, T, caption_len=32, NT=4):
    """v1 TVTS: tubelet embedding on kept patches, joint attention over S = 1 + tubes * n tokens, DistilBERT
    at the padded caption length, sorting head on the 768-wide tokens; no dgrad for the patch embedding."""
    p, W, E, Wt, Ws = a["patch"], a["width"], a["embed"], a["text_width"], a["sort_width"]
    tubes = T // a["tubelet"]
    n = int((a["image"] // p) ** 2 * (1 - a["mask_ratio"]))
    S, L, Lt = 1 + tubes * n, caption_len, a["text_layers"]
    So = S + NT
    patch = 2 * tubes * n * 3 * a["tubelet"] * p * p * W
    sort = (2 * So * 24 * Ws * Ws + 8 * So * So * Ws + 2 * NT * Ws * a["n_trans"]) if NT > 1 else 0
    fwd = (patch + a["layers"] * (S * 24 * W * W + 4 * S * S * W) + 2 * W * E
           + NT * (Lt * (L * (8 * Wt * Wt + 4 * Wt * a["text_ffn"]) + 4 * L * L * Wt) + 2 * Wt * E) + sort)
    return fwd, 2 * fwd - patch


def cpu_baseline_worker(arch_name, T, caption_len, pairs, max_seconds, threads, n_trans=4):
    """The CPU oracle (oracle/tvts_oracle.py, a port) timed on this host: full step incl. HF-AdamW."""
    from oracle import tvts_oracle as O
    torch.set_num_threads(threads)
    if arch_name == "v1":
        from oracle import tvts_v1_oracle as V
        oarch = V.ARCH
        P = V.synth_params(oarch, seed=0)
        batch = V.synth_batch(oarch, B=pairs, T=T, seed=0, caption_len=caption_len, n_trans=n_trans)
        mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in P.items()}

        class _O:  # the v1 step: one AdamW group, lr 1e-4, weight_decay 0 (v1/configs/dist-yt-pt.json)
            @staticmethod
            def train_step(P, batch, oarch, state):
                state["t"] = state.get("t", 0) + 1
                leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
                l1, l2, *_ = V.step_losses(leaves, batch, oarch)
                (l1 + l2).backward()
                for k in P:
                    g = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(P[k])
                    O.hf_adamw_step(P[k], g, mom[k][0], mom[k][1], state["t"], 1e-4, 0.0)
        O = _O  # noqa: N806
    else:
        oarch = O.ARCHS[arch_name]
        P = O.synth_params(oarch, seed=0)
        batch = O.synth_batch(oarch, B=pairs, T=T, seed=0, caption_len=caption_len, n_trans=n_trans)
    # a short thread sweep ON THIS HOST picks the thread count (one step each at half the timed batch, after one warm-up step): the
    # oracle's CPU step is memory-bound and gets SLOWER with too many torch threads, and the best count depends on the host
    host = host_threads()
    cands = sorted({max(1, threads // 2), threads, min(host, threads * 2)})
    sweep = {}
    if len(cands) > 1 and max_seconds >= 10:
        if arch_name == "v1":
            small = V.synth_batch(oarch, B=max(1, pairs // 2), T=T, seed=1, caption_len=caption_len, n_trans=n_trans)
        else:
            small = O.synth_batch(oarch, B=max(1, pairs // 2), T=T, seed=1, caption_len=caption_len, n_trans=n_trans)
        Ps, st = {k: v.clone() for k, v in P.items()}, {}
        torch.set_num_threads(threads)
        O.train_step(Ps, small, oarch, st)  # allocator / thread-pool warm-up
        for c in cands:
            torch.set_num_threads(c)
            t0 = time.time()
            O.train_step(Ps, small, oarch, st)
            sweep[c] = max(1, pairs // 2) / (time.time() - t0)
            if time.time() - t0 > max_seconds / 2:
                break  # a count that slow is not going to win; keep the sweep bounded
        threads = max(sweep, key=sweep.get)
        del Ps, st, small
    torch.set_num_threads(threads)
    state = {}
    t0 = time.time()
    O.train_step(P, batch, oarch, state)  # first step (includes allocator warm-up)
    first = time.time() - t0
    n, dt = 1, first
    if first < max_seconds / 2:
        t0, n = time.time(), 0
        while n < 1 or (time.time() - t0 + first < max_seconds and n < 4):
            O.train_step(P, batch, oarch, state)
            n += 1
        dt = (time.time() - t0) / n
    swept = ("thread sweep on this host (pairs/s at %d pairs per step): " % max(1, pairs // 2)
             + ", ".join(f"{c} threads {v:.2f}" for c, v in sweep.items()) + f" -> {threads}") if sweep else "no thread sweep"
    return {"value": pairs / dt, "unit": "pairs/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "thread_sweep": {str(c): v for c, v in sweep.items()},
            "sample": f"{n} full step(s) (fwd+bwd+HF-AdamW) of the fp32 torch-CPU oracle, {arch_name}, T={T}, "
                      f"{pairs} pairs/step, {caption_len}-token captions x{n_trans}, {threads} torch threads on a host with "
                      f"{os.cpu_count()} logical CPUs ({host} usable by this process); {swept}"}


def host_threads():
    """threads for the CPU baseline: every CPU this process may run on (cgroup / affinity aware)"""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(arch_name, T, caption_len, pairs, max_seconds, n_trans=4):
    """Runs the worker in a child process with a hard wall-clock limit, so a slow host cannot stall the bench."""
    import subprocess
    # starting point of the worker's own thread sweep (it also tries half and twice this count): 32 was best on the 256-CPU hosts
    # of rounds 2-4 (0.98 pairs/s; 64 threads 0.52, 128 threads 0.25, 256 timed out), fewer CPUs -> all of them
    threads = min(host_threads(), 32)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--arch", arch_name, "--frames", str(T),
           "--caption-len", str(caption_len), "--cpu-pairs", str(pairs), "--cpu-seconds", str(max_seconds),
           "--cpu-threads", str(threads), "--n-trans", str(n_trans)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=4 * max_seconds + 60)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout or failure: the GPU number still stands
        return {"value": None, "unit": "pairs/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
                "sample": f"CPU oracle step did not finish within {4 * max_seconds + 60:.0f} s ({type(e).__name__})"}


def gemm_bytes(kind, shp):
    """operand + result bytes of one GEMM launch if every matrix crossed HBM exactly once (the floor `traffic` is read
    against)."""
    if kind in ("gemm_tn", "gemm_tn_fp8"):
        M, Na, Nb = shp[:3]
        return (1.0 if kind.endswith("fp8") else 2.0) * M * (Na + Nb) + 4.0 * Na * Nb
    M, N, K, odt, res, act, gate = shp
    eb = 1.0 if odt == "fp8" else 2.0
    b = eb * K * (M + N) + M * N * (4.0 if odt == "f32" else 2.0)
    if res:
        b += 4.0 * M * N
    if act:      # pre-activation side output kept for the backward
        b += 2.0 * M * N
    if gate:     # activation-gradient gate reads the saved pre-activation
        b += 2.0 * M * N
    return b


GEMM_SOURCES = ("gemm.hip", "gemm_nt256.h", "gemm_nt_ring.h", "gemm_tn256.h", "gemm_tn8.h", "common.h")


def gemm_source_id():
    """sha256[:16] over the GEMM kernel sources: the build id a committed PMC traffic measurement is stamped with"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tvts_amd", "csrc")
    for f in GEMM_SOURCES:
        h.update(open(os.path.join(base, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(args):
    """HBM bytes of the step's GEMM launches from a committed PMC pass of this exact workload (counters cannot be read
    live: they need rocprofv3's own passes, tools/pmc_traffic.sh).  The file carries the id of the GEMM sources it was measured
    on (gemm_source_id); a file measured on other sources is REFUSED (returns a dict with only "stale"), so a kernel change cannot
    keep quoting an old number.  None when no measurement of this workload is committed."""
    import glob
    if args.fp8 or args.n_trans != 4:
        return None
    pat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                       f"*pmc_step_traffic_{args.arch}_t{args.frames}_b{args.batch}.json")
    hits = sorted(glob.glob(pat))
    if not hits:
        return None
    d = json.load(open(hits[-1]))
    d["source"] = "profiles/" + os.path.basename(hits[-1])
    if d.get("gemm_source_id") != gemm_source_id():
        return {"stale": f"{d['source']} was measured on GEMM sources {d.get('gemm_source_id', '(unstamped)')}, this build is "
                         f"{gemm_source_id()}: re-run tools/pmc_traffic.sh"}
    return d


def pmc_mfma_util(args):
    """Matrix-pipe duty and clock under load of the step's GEMM launches from a committed counter pass of this exact workload
    (tools/pmc_mfma_util.sh: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE; counters need rocprofv3's own passes).  Stamped and refused
    like pmc_traffic."""
    import glob
    if args.fp8 or args.n_trans != 4:
        return None
    pat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                       f"*pmc_mfma_util_{args.arch}_t{args.frames}_b{args.batch}.json")
    hits = sorted(glob.glob(pat))
    if not hits:
        return None
    d = json.load(open(hits[-1]))
    src = "profiles/" + os.path.basename(hits[-1])
    if d.get("gemm_source_id") != gemm_source_id():
        return {"stale": f"{src} was measured on GEMM sources {d.get('gemm_source_id', '(unstamped)')}, this build is "
                         f"{gemm_source_id()}: re-run tools/pmc_mfma_util.sh"}
    a = d["all_gemm"]
    return {"source": src, "mfma_busy": a["mfma_busy"], "clock_mhz_under_load": 1e3 * a["clock_ghz"],
            "peak_at_clock": a["peak_tflops_at_clock"], "tflops_in_profiled_pass": a["tflops"],
            "by_family": {k: {"mfma_busy": v["mfma_busy"], "clock_mhz": 1e3 * v["clock_ghz"]} for k, v in d.get("by_family", {}).items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="B_16")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--batch", type=int, default=192, help="pairs per GPU (the reference config uses 12 on V100)")
    ap.add_argument("--caption-len", type=int, default=32)
    ap.add_argument("--dense-sort-head", action="store_true", help="evaluate the last block of the sort head and of the text tower on every row like "
                    "the reference does (default: on the rows the model reads -- NT transcript rows / EOT row; same loss and gradients)")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 4: e4m3 forward GEMMs in the ViT blocks")
    ap.add_argument("--fp8-dgrad", action="store_true", help="... and e4m3 input-gradient GEMMs (output gradient one scale per token, "
                    "transposed e4m3 weight copies); the weight gradients stay bf16")
    ap.add_argument("--bf16-grad-stream", action="store_true", help="carry the GRADIENT of the ViT blocks' residual stream in bf16 on EVERY row, the "
                    "forward stream in fp32 (round 3's default: 2.4x the error of the embedding-side gradients; switches the hybrid stream off)")
    ap.add_argument("--fp8-wgrad", action="store_true", help="BASELINE config 5 in full: e4m3 weight gradients as well (implies --fp8-dgrad); every "
                    "e4m3 operand copy under one delayed scale per tensor, tvts_gemm_tn_fp8")
    ap.add_argument("--fp32-streams", action="store_true", help="the space-time blocks' residual stream and its gradient in fp32 on every row "
                    "(rounds 1, 2, 4) instead of the hybrid stream (round 5 default: bf16 rows, the CLS row of every clip in fp32)")
    ap.add_argument("--bf16-residual", action="store_true", help="both streams in bf16 on EVERY row, CLS rows included (round 3's opt-in; "
                    "the |d loss| <= 1e-2 gate of SURVEY 8d fails on one small configuration; switches the hybrid stream off)")
    ap.add_argument("--wgrad-stream", choices=("auto", "on", "off"), default="auto", help="weight gradients of the ViT blocks on a side "
                    "stream beside the input-gradient chain (auto = off: measured slower, profiles/r04_wgrad_side_stream.txt)")
    ap.add_argument("--tn-grouped", choices=("auto", "on", "off"), default="auto", help="the six weight gradients of a ViT block in one grouped "
                    "launch (auto: from 6 000 to 12 000 token rows per GPU, Engine.TN_GROUPED_MIN_ROWS / MAX_ROWS -- the reference's 12 pairs; "
                    "profiles/r04_tn_grouped.txt)")
    ap.add_argument("--fp8-q8-only", choices=("on", "off"), default="on", help="with --fp8-wgrad: the MLP's GELU output and gated hidden gradient "
                    "leave their GEMM epilogues as e4m3 bytes ONLY (round 5 default); off = bf16 result + a quantiser pass (round 4)")
    ap.add_argument("--text-side", choices=("on", "off"), default="on", help="the text tower on its own stream beside the ViT (round 5 default; "
                    "off = in line in front of it)")
    ap.add_argument("--graph", action="store_true", help="world 1: capture the whole step into a hipGraph per resident batch and replay it.  Since round "
                    "6 the default is EAGER launches wherever the step is not launch-bound: with the text tower on its own stream the replayed "
                    "two-stream graph measures SLOWER than the eager launches of the same step -- 0.6 - 0.9 %% at 192 pairs, 4.5 %% at 24, 6.5 %% at 12 "
                    "(profiles/r06_bench_product_path.txt) -- and the eager step is the trainer's own path.  Where the step is launch-bound (GRAPH_AUTO_WORK: up to 4 pairs of "
                    "ViT-B/16, ~800 launches for 9 ms of kernels) the host cannot keep up and the replay is taken automatically")
    ap.add_argument("--no-graph", action="store_true", help="eager launches whatever the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pairs", type=int, default=8, help="pairs per step of the CPU baseline (the AdamW pass over 156 M "
                    "parameters is amortised over them, as it is over the GPU batch)")
    ap.add_argument("--n-trans", type=int, default=4, help="captions per video: 4 = YT-Temporal batch with the sorting head, "
                    "1 = WebVid-style batch (no sorting head / loss, SURVEY.md 8d)")
    ap.add_argument("--cpu-threads", type=int, default=8)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_worker(args.arch, args.frames, args.caption_len, args.cpu_pairs, args.cpu_seconds,
                                             args.cpu_threads, args.n_trans)), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test hooks for the single-GPU box (tests/test_bench_dist_gpu.py): every rank on device 0 with the gloo backend --
    # the collectives' call pattern is what is being checked there; the real run is one rank per GPU over RCCL
    if os.environ.get("TVTS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("TVTS_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from tvts_amd import arch as A
    from tvts_amd import hip as K
    from tvts_amd.data_loader import synth_batch, synth_batch_v1
    from tvts_amd.model._common import TVTSv2Base
    from tvts_amd.model.model_dist_TVTS import TVTS
    from tvts_amd.optim import FusedHFAdamW
    from tvts_amd.step import StepRunner

    a = dict(A.ARCHS[args.arch])
    if args.frames > a["num_frames"]:  # BASELINE config 3: 16-frame clips need a temporal table past the reference's 12 rows
        a["num_frames"] = args.frames
    if args.fp8_wgrad:
        args.fp8_dgrad = True
        a["fp8_wgrad"] = True
    if args.fp8_dgrad:
        args.fp8 = True
        a["fp8_dgrad"] = True
    if args.fp8:
        a["fp8"] = True
    if args.dense_sort_head:
        a["sort_used_rows_only"] = False
    if args.fp32_streams:
        a["hybrid_stream"] = False
    if args.bf16_grad_stream:
        a["bf16_grad_stream"] = True
    if args.bf16_residual:
        a["bf16_residual"] = True
    a["text_side"] = args.text_side == "on"
    if args.fp8_q8_only == "off":
        a["fp8_q8_only"] = False
    if args.tn_grouped != "auto":
        a["tn_grouped"] = args.tn_grouped == "on"
    if os.environ.get("TVTS_TN_GROUP_SPLITS"):
        a["tn_group_splits"] = int(os.environ["TVTS_TN_GROUP_SPLITS"])
    if args.wgrad_stream != "auto":
        a["wgrad_stream"] = args.wgrad_stream == "on"
    margs = types.SimpleNamespace(local_rank=local_rank, rank=rank, world_size=world)
    v1 = a.get("family") == "v1"
    if v1:
        synth_batch = synth_batch_v1
        hp = ((1e-4, 0.0),) * 4  # one parameter group in the v1 entrypoint (configs/dist-yt-pt.json)
    else:
        hp = A.GROUP_HPARAMS
    model = (TVTS if v1 else TVTSv2Base)(margs, arch=a, init_seed=0)
    groups = [[], [], [], []]
    for name, p in model.named_parameters():
        gi = A.param_group_of(name, a)
        if gi < 0:
            p.requires_grad = False
        else:
            groups[gi].append(p)
    opt = FusedHFAdamW([dict(params=groups[i], lr=hp[i][0], weight_decay=hp[i][1]) for i in range(4) if groups[i]],
                       model.store, model=model)
    runner = StepRunner(model, opt)
    B, T = args.batch, args.frames
    dev = model.store.device

    # two resident synthetic batches (rank-dependent seeds), prepared once and used alternately: the timed step starts
    # from inputs that already live in HBM, with no staging copy
    pool = [synth_batch(a, B, T, seed=1000 * rank + i, caption_len=args.caption_len, n_trans=args.n_trans) for i in range(2)]
    model._fresh_shadows(); model._sync_requires_grad()
    pbs = [model.engine.prepare_batch(b) for b in pool]
    labels = [b["label"].reshape(-1).to(torch.int32).to(dev) if "label" in b else None for b in pool]
    srcs = pbs

    def one_step(i, device_step):
        return runner.run(pbs[i % len(pbs)], labels[i % len(pbs)], device_step=device_step)

    # Eager launches by default (round 6: the same StepRunner.run the trainer drives; the device is never short of queued work -- 800
    # launches of ~4 us host time per 15 ... 138 ms step).  --graph (world 1): the whole step (zero_grad ... AdamW) captured once per
    # resident batch and replayed -- measured slower than eager since the text tower runs on its own stream (see the flag's help).
    # world > 1: RCCL calls are capturable and the native transport's fork / join events make the side stream part of the capture
    # (tests/test_comm_gpu.py::test_captured_step_through_the_native_transport does it at world 1), but no multi-GPU box was
    # available to validate a captured multi-rank step: TVTS_BENCH_GRAPH_DDP=1 + --graph.
    from tvts_amd import dist as D
    launch_bound = (not v1) and float(B * (1 + T * A.n_keep(a))) * a["width"] ** 2 < GRAPH_AUTO_WORK   # ~800 launches on a handful of clips
    use_graph = (args.graph or (launch_bound and world == 1)) and not args.no_graph and (
        (world == 1) or (os.environ.get("TVTS_BENCH_GRAPH_DDP") == "1" and D.transport() == "native"))
    for i in range(max(args.warmup, 1)):
        out = one_step(i, device_step=use_graph)
    torch.cuda.synchronize()
    graphs = None
    if use_graph:
        try:
            graphs = []
            for gi in range(len(srcs)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = one_step(gi, device_step=True)
                graphs.append(g)
            torch.cuda.synchronize()
        except Exception as e:  # stay correct: fall back to eager launches
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs, use_graph = None, False
            torch.cuda.synchronize()

    if world > 1 and graphs is None:
        runner.diag = {}  # exchange diagnostics of the timed steps: where the compute stream waits for a collective (StepRunner._bracket)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if graphs is not None:
            graphs[i % len(graphs)].replay()
        else:
            out = one_step(i, device_step=False)
    torch.cuda.synchronize()
    # this rank's own time for the K steps (before the closing barrier): the per-rank spread tells a slow rank from a slow collective
    t_own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    diag, runner.diag = runner.diag, None
    exch_diag = {}
    if world > 1:
        if diag is not None:
            mine = [sum(a.elapsed_time(b) for a, b in diag.get(k, [])) / max(args.steps, 1) for k in ("gather", "allreduce")]
        else:
            mine = [float("nan"), float("nan")]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank = torch.zeros(world, 3, dtype=torch.float64, device=dev)
        per_rank[rank, 0], per_rank[rank, 1], per_rank[rank, 2] = 1e3 * t_own / args.steps, mine[0], mine[1]
        dist.all_reduce(per_rank)
        per_rank = per_rank.cpu()
        dt = float(t[0])
        exch_diag = {
            # ms per step the compute stream spent waiting where the loss needs the gathered embeddings / where AdamW needs the reduced
            # gradient (HIP event pairs around EmbedGather.result / GradSync.finish in every timed step; mean over ranks, and the worst)
            "gather_wait_ms": float(per_rank[:, 1].mean()), "gather_wait_ms_max_rank": float(per_rank[:, 1].max()),
            "allreduce_exposed_ms": float(per_rank[:, 2].mean()), "allreduce_exposed_ms_max_rank": float(per_rank[:, 2].max()),
            "step_ms_per_rank": [float(x) for x in per_rank[:, 0]],
            "step_ms_rank_spread": float(per_rank[:, 0].max() - per_rank[:, 0].min()),
            "diag": "HIP events on the compute stream, eager steps" if diag is not None else "not collected (captured multi-rank step)",
            "cu_reservation": dict(D.CU_RESERVATION),
            "text_range": "handed to the all-reduce at the join, behind the ViT ranges" if model.engine.text_side else "first (text tower in line)",
        }
    loss = float(out["loss1"]) + (float(out["loss2"]) if out["loss2"] is not None else 0.0)

    fwd, bwd = (step_flops_per_pair_v1 if v1 else step_flops_per_pair)(a, T, args.caption_len, args.n_trans)
    # matmul FLOPs the step does NOT execute when the sort head's last block runs on the rows the model reads (engine.sort_forward):
    # attention output / projection / MLP of the Sv video rows of that block, forward and backward (its qkv projection stays dense)
    skipped = 0.0
    if args.n_trans > 1 and args.n_trans <= 16 and not args.dense_sort_head:
        Es = a.get("sort_width", a["embed"])
        So_ = int(pbs[0]["S"]) - (1 if a["tail"] == "pooled_and_patches" else 0) + args.n_trans
        skipped = 3.0 * (So_ - args.n_trans) * (18 * Es * Es + 4 * So_ * Es)
    if not args.dense_sort_head and not v1:  # ... and the text tower's last block on the EOT row of every caption
        Wt_, L_ = a["text_width"], args.caption_len
        skipped += 3.0 * args.n_trans * (L_ - 1) * (18 * Wt_ * Wt_ + 4 * L_ * Wt_)
    # proof of participation for a SCALE record: every rank adds 1 (and its device index) through the step's own transport path
    ranks_seen, devices_seen = 1, [torch.cuda.current_device()]
    if world > 1:
        probe = torch.zeros(world + 1, dtype=torch.float32, device=dev)
        probe[0] = 1.0
        probe[1 + rank] = float(torch.cuda.current_device()) + 1.0
        dist.all_reduce(probe)
        ranks_seen = int(round(float(probe[0])))
        devices_seen = [int(round(float(x))) - 1 for x in probe[1:].tolist()]
    native_ok = None
    if D.transport() == "native":
        try:
            nc = D.NativeComm.get()
            nc.self_test()
            native_ok = {"self_test": "ok", "comm_world": int(nc.W), "comm_rank": int(nc.rank)}
        except Exception as e:
            native_ok = {"self_test": f"failed: {type(e).__name__}: {e}"}
    pairs_per_s = world * B * args.steps / dt
    line = {
        "metric": "video-text pairs/sec/node, TVTSv2 pretrain step", "value": pairs_per_s, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("fp8 e4m3 forward" + (" + input-gradient" if args.fp8_dgrad else "") + (" + weight-gradient" if args.fp8_wgrad else "") + " GEMMs (ViT blocks, v_mfma_scale_f32_16x16x128_f8f6f4) + bf16") if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": (f"TVTS v1 ViT-B/16 tubelet 2, {T}-frame 224^2, mask {a['mask_ratio']}, DistilBERT, " if v1 else
                                f"TVTSv2 ViT-{args.arch.replace('_', '/')} {T}-frame 224^2, mask {a['mask_ratio']}, ")
                               + f"{args.caption_len}-token captions x{args.n_trans}, full pretrain step (fwd+losses+bwd+HF-AdamW)",
                   "pairs_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}",
                   "hip_graph": bool(graphs is not None), "step_gflop_per_pair": (fwd + bwd) / 1e9,
                   "executed_gflop_per_pair": (fwd + bwd - skipped) / 1e9,
                   "last_blocks": "dense" if args.dense_sort_head else "sort head / text tower last block on the rows the model reads (NT transcript rows / EOT row)",
                   "final_loss": loss,
                   # peak of the caching allocator over the whole process (parameters + optimizer state + every saved activation + workspaces):
                   # the "288 GB HBM sizing" figure of BASELINE configs[3]
                   "hbm_peak_gb": torch.cuda.max_memory_allocated() / 1e9, "hbm_reserved_gb": torch.cuda.max_memory_reserved() / 1e9,
                   "exchange": {"transport": D.transport() + (f" ({backend})" if world > 1 else ""),
                                "ranks_seen": ranks_seen, "devices_seen": devices_seen, "native": native_ok,
                                "grad_payload": runner.sync.payload, "grad_bytes_per_step": runner.sync.bytes_sent, **exch_diag}},
        "step_mfma_frac": pairs_per_s * (fwd + bwd - skipped) / (world * PEAK_BF16_TFLOPS * 1e12),
    }

    if not args.no_roofline:
        # dominant kernel family = the bf16 MFMA GEMMs (gemm_nt_kernel / gemm_tn_kernel): one instrumented eager
        # step with a HIP event pair around every GEMM launch on the launch stream.  EVERY rank runs the step (it
        # contains the all-gather / all-reduce of the data-parallel path); rank 0 records.
        # The text tower runs IN LINE in this one step (in the timed steps it runs on its own stream beside the ViT, same kernels,
        # same bits): a launch's event pair then brackets that launch alone, not the other stream's kernels it shared the chip with.
        if rank == 0:
            K.GEMM_PROFILE = []
            K.HBM_PROFILE = []
            # the shader clock the chip holds under this load, sampled INSIDE the plain bf16 launches of the 256 x 256 kernel (qkv forward,
            # input gradients): block 0 reads s_memtime (shader cycles) and s_memrealtime (constant rate) at its start and end
            K.CLOCK_PROBE = dict(buf=torch.zeros(256, 4, dtype=torch.int64, device=dev), i=0, shape=[])
        eng_ts, model.engine.text_side = model.engine.text_side, False
        # (a few un-instrumented steps first: the power management settles on the clock of the sustained step, not of the pause behind the timed loop)
        for i in range(2):
            one_step(i, device_step=False)
        if rank == 0:
            K.GEMM_PROFILE, K.HBM_PROFILE = [], []
            K.CLOCK_PROBE.update(i=0, shape=[])
        one_step(0, device_step=False)
        torch.cuda.synchronize()
        model.engine.text_side = eng_ts
        if world > 1:
            dist.barrier()
    if rank == 0 and not args.no_roofline:
        recs, K.GEMM_PROFILE = K.GEMM_PROFILE, None
        hrecs, K.HBM_PROFILE = K.HBM_PROFILE, None
        cp, K.CLOCK_PROBE = K.CLOCK_PROBE, None
        wall_khz, n_cus, sheet_khz = K.device_clock_info(torch.cuda.current_device())
        pr = cp["buf"][:cp["i"]].cpu().double()
        cyc, tk = pr[:, 2] - pr[:, 0], pr[:, 3] - pr[:, 1]
        ok = (pr[:, 0] > 0) & (pr[:, 2] > 0) & (tk > 0)          # launches that took the sampling instantiation
        clk = (cyc[ok] / tk[ok] * wall_khz / 1e3) if bool(ok.any()) else None   # MHz per sampled launch
        clk_w = tk[ok] if clk is not None else None               # weights: the launches' durations
        tot_ms = sum(r[2].elapsed_ms(r[3]) for r in recs)
        tot_fl = sum(r[1] for r in recs)
        by = {}
        shapes = {}
        for kind, f, s, e, shp in recs:
            d = by.setdefault(kind, [0, 0.0, 0.0]); d[0] += 1; d[1] += f; d[2] += s.elapsed_ms(e)
            d2 = shapes.setdefault((kind,) + tuple(shp), [0, 0.0, 0.0]); d2[0] += 1; d2[1] += f; d2[2] += s.elapsed_ms(e)
        if os.environ.get('TVTS_BENCH_SHAPES'):
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][2])[:24]:
                print(f'[shape] {v[2]:7.2f} ms n={v[0]:3d} {v[1] / (v[2] * 1e-3) / 1e12:7.1f} TF  {k}', file=sys.stderr)
        if os.environ.get('TVTS_BENCH_ORDER'):   # dev: launch-ordered GEMM list (joined with PMC rows by tools/pmc_join.py)
            json.dump([[kind, list(shp), s.elapsed_ms(e), gemm_bytes(kind, shp)] for kind, f, s, e, shp in recs],
                      open(os.environ['TVTS_BENCH_ORDER'], 'w'))
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        traffic = pmc_traffic(args)
        stale = traffic.get("stale") if traffic else None
        if stale:
            traffic = None
        # every launch priced against the dense peak of ITS operand type (bf16 2.5, e4m3 through the K = 128 scaled MFMA 5 PFLOP/s): the
        # fraction is the time the launches would take at their own peaks over the time they took (= achieved / peak on a pure bf16 step)
        ideal_ms = sum(v[1] / ((PEAK_FP8_TFLOPS if k.endswith("fp8") else PEAK_BF16_TFLOPS) * 1e12) * 1e3 for k, v in by.items())
        line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": ideal_ms / tot_ms,
                            "frac_of": "the MFMA GEMM launches alone (summed 2MNK over summed launch durations); the WHOLE step on its "
                                       "executed FLOPs is step_frac (= step_mfma_frac)",
                            # the fraction BASELINE.md section 3 defines: executed matmul FLOPs of the whole step (every kernel's
                            # time in the denominator, not only the GEMMs') over the dense bf16 peak, from the TIMED replayed steps
                            "step_frac": line["step_mfma_frac"],
                            "traffic": traffic["hbm_bytes_per_step"] if traffic else None,
                            "traffic_unit": "HBM bytes per step over the same launches (rocprofv3 --pmc FETCH_SIZE x2 + "
                                            "WRITE_SIZE, tools/pmc_traffic.sh)" if traffic else None,
                            "traffic_source": traffic["source"] if traffic else stale,
                            "algorithmic_bytes": sum(gemm_bytes(r[0], r[4]) for r in recs),
                            "kernel": "gemm_nt_kernel + gemm_tn_kernel (all MFMA GEMM launches of one step)",
                            "launches": len(recs), "gemm_ms_per_step": tot_ms,
                            # the e4m3 launches (--fp8) run on the K = 128 scaled MFMA and are priced against the 5 PFLOP/s dense fp8 peak
                            "by_kernel": {k: {"launches": v[0], "tflops": v[1] / (v[2] * 1e-3) / 1e12, "ms": v[2],
                                              "peak": PEAK_FP8_TFLOPS if k.endswith("fp8") else PEAK_BF16_TFLOPS,
                                              "frac": v[1] / (v[2] * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if k.endswith("fp8") else PEAK_BF16_TFLOPS)}
                                          for k, v in by.items()}}
        # matrix-pipe duty and the clock the GEMM launches actually ran at ("confirm the peak on the box"): a committed
        # counter pass of this workload; frac_at_clock prices the live `achieved` against the dense peak AT THAT CLOCK
        # (256 CUs x 4 SIMDs x 1024 FLOP per cycle) beside the sheet's 2.5 PFLOP/s at 2.4 GHz
        util = pmc_mfma_util(args)
        if clk is not None:
            # LIVE: the in-kernel clock samples of this very step; the peak at that clock = CUs x 4 SIMDs x 1024 FLOP per cycle (v_mfma_f32_16x16x32_bf16:
            # 16 384 FLOP in 16 cycles per SIMD; 2.5 PFLOP/s is 256 CUs at the sheet's 2.4 GHz)
            mhz = float((clk * clk_w).sum() / clk_w.sum())
            peak_clk = n_cus * 4 * 1024 * mhz * 1e6 / 1e12
            line["roofline"].update({"clock_mhz_under_load": mhz, "clock_mhz_min_max": [float(clk.min()), float(clk.max())],
                                     "clock_samples": int(len(clk)), "sheet_clock_mhz": sheet_khz / 1e3, "cu_count": n_cus,
                                     "peak_at_clock": peak_clk, "frac_at_clock": ach / peak_clk,
                                     "clock_source": "shader cycles (s_memtime) over constant-rate ticks (s_memrealtime) sampled by block 0 INSIDE the "
                                                     "plain bf16 256 x 256 NT launches of the instrumented step (TVTS_GEMM_CLOCK_SAMPLE), duration-weighted"})
        if util and "stale" not in util:
            line["roofline"].update({"mfma_busy": util["mfma_busy"], "mfma_busy_clock_mhz": util["clock_mhz_under_load"],
                                     "mfma_util_by_family": util["by_family"], "mfma_util_source": util["source"]})
        else:
            line["roofline"].update({"mfma_busy": None, "mfma_util_source": util["stale"] if util else None})
        # the HBM-bound families of the same instrumented step (LayerNorm, attention, AdamW, weight transposes): algorithmic bytes
        # of every launch (each operand and result once) over its HIP-event duration, against the 8 TB/s peak of
        # MI355X_MICROARCH.md and against the ~6.3 TB/s a streaming kernel reaches on this part
        fam = {}
        for name, nbytes, e0, e1 in hrecs:
            d = fam.setdefault(name, [0, 0.0, 0.0]); d[0] += 1; d[1] += nbytes; d[2] += e0.elapsed_ms(e1)
        step_ms = 1e3 * dt / args.steps
        line["roofline"]["hbm_kernels"] = {
            k: {"launches": v[0], "ms": v[2], "share_of_step": v[2] / step_ms, "algorithmic_GB": v[1] / 1e9,
                "GBps": v[1] / (v[2] * 1e-3) / 1e9 if v[2] > 0 else None,
                "frac_of_peak_8TBps": v[1] / (v[2] * 1e-3) / PEAK_HBM_BPS if v[2] > 0 else None,
                "frac_of_achievable_6.3TBps": v[1] / (v[2] * 1e-3) / ACHIEVABLE_HBM_BPS if v[2] > 0 else None}
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])}
        line["roofline"]["hbm_kernels_note"] = ("HIP events around every launch of one eager step; attention entries include their "
                                                "CLS merge / delta / finalize kernels; small-row LayerNorms (text / sort-head rows) apart")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.arch, T, args.caption_len, pairs=args.cpu_pairs, max_seconds=args.cpu_seconds,
                                            n_trans=args.n_trans)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
