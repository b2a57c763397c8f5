#!/usr/bin/env python3
"""Writes profiles/INDEX.md: claim -> the CURRENT file -> the superseded files of earlier rounds (same measurement, older tree).
Files are named r<round>_<what>; the current file of a measurement is the one of the highest round.  usage: python tools/profiles_index.py"""
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")

# what -> (claim it supports, where DESIGN.md quotes it)
RECURRING = collections.OrderedDict([
    ("bench_default_driver_form_final_tree.json", ("the driver-form line once more on the round's last commit (another box)", "")),
    ("bench_default_driver_form.json", ("THE HEADLINE: `python bench.py` exactly as the driver runs it (pairs/s, `roofline` with clock / duty / traffic, `cpu_baseline`)", "section 7, Headline")),
    ("bench_default_b192.json", ("default workload, 20 timed steps, without the CPU leg", "section 7")),
    ("bench_default_b192_graph_replay.json", ("the same with `--graph` (captured two-stream step replayed): slower than eager", "section 7, bench.py contract")),
    ("bench_dense_sort_head.json", ("`--dense-sort-head`: last block of the sort head / text tower on every row, as the reference evaluates them", "section 0, row A8")),
    ("kernel_summary_default_b192.txt", ("per-kernel time shares of the 192-pair step (`rocprofv3 --kernel-trace`, eager, text tower in line)", "section 7")),
    ("kernel_stats_default_b192.csv", ("the same as CSV (every kernel)", "section 7")),
    ("pmc_step_traffic_B_16_t8_b192.json", ("`roofline.traffic`: HBM-side bytes of the GEMM launches of one step (FETCH_SIZE x 2 + WRITE_SIZE), stamped with the GEMM source hash", "section 7, Traffic")),
    ("pmc_gemm_traffic_by_shape.txt", ("the same joined per GEMM shape against the algorithmic bytes", "section 7, Traffic")),
    ("pmc_mfma_util_B_16_t8_b192.txt", ("matrix-pipe duty, wait / issue-stall split and clock per GEMM shape and per attention / LayerNorm kernel", "section 7, power-limited")),
    ("pmc_mfma_util_B_16_t8_b192.json", ("the same as JSON: `roofline.mfma_busy` of the bench line (stamped)", "section 7")),
    ("mfma_power_probe.txt", ("what a pure MFMA stream sustains on this part: shape (16x16x32 vs 32x32x16) x operands (random vs zero) x LDS reads; clock per arm", "section 7, power-limited")),
    ("cu_store_probe.txt", ("what ONE CU can store (64 B/clk) against all 256 at once (10 - 17 B/clk = the chip's write bandwidth): the NT epilogue's store tail is not the CU's store path", "section 7, Where the GEMM time is")),
    ("gemm_trace_wide_forms.txt", ("per-tile trace of the wide NT forms (fc1 forward, fc2 input gradient, qkv forward): epilogue length against the number of blocks inside an epilogue at the same time -- no dependence: the tail is CU-local", "section 7, Where the GEMM time is")),
    ("gemm_ab_b16_patch.txt", ("bf16-first LDS patch against the fp32 patch per GEMM shape (adopted for the plain forms: same bits, 3 - 5 % per launch)", "section 5")),
    ("gemm_ab_b16_side_input_forms.txt", ("the bf16-first patch for the forms with a bf16 side input (side values loaded in the accumulator layout): same bits, 2 - 7 % SLOWER -- not adopted", "section 9, item 2")),
    ("bench_b16_patch_ab.txt", ("the step with / without the bf16-first patch, alternating runs on one box: +0.75 %", "section 5")),
    ("bench_product_path.txt", ("product path against bench path at 192 / 24 / 12 pairs: eager vs `--graph`, host-fed step, trainer epoch (eager and `TVTS_TRAINER_GRAPH=1`)", "section 7, table")),
    ("bench_launch_bound_eager_vs_graph.txt", ("eager against replayed graph where the step is launch-bound: 2 / 4 / 6 pairs of B/16, H/14 at its 2 pairs of 16 frames (the rule behind bench.py's automatic --graph)", "section 7, Other configurations")),
    ("bench_reference_batches.jsonl", ("the reference's own per-GPU batches (2 / 12 / 24 pairs) and WebVid-style NT = 1", "section 7")),
    ("bench_b32_t8.jsonl", ("BASELINE configs[1]: ViT-B/32, 8 frames, 384 and 24 pairs", "section 7")),
    ("bench_h14_t16_b48.jsonl", ("BASELINE configs[3] / [4]: H/14, 16 frames, 48 pairs: bf16, --fp8, --fp8-dgrad, --fp8-wgrad back to back", "section 8")),
    ("bench_h14_max_batch.jsonl", ("H/14 16-frame sizing: 64 / 80 / 96 pairs with `config.hbm_peak_gb` (96 pairs = 267.7 GB)", "sections 4, 7")),
    ("bench_v1.jsonl", ("v1 TVTS step (row N4)", "section 7")),
    ("kernel_summary_h14_b48_fp8_wgrad.txt", ("kernel shares of the e4m3 H/14 step", "section 8")),
    ("kernel_summary_h14_b48_bf16.txt", ("kernel shares of the bf16 H/14 step", "section 8")),
    ("kernel_summary_v1_t4_b256.txt", ("kernel shares of the v1 step", "")),
    ("kernel_summary_b12.txt", ("kernel shares of the 12-pair step", "section 9, item 4")),
    ("kernel_summary_b24.txt", ("kernel shares of the 24-pair step", "")),
    ("timeline_b12.txt", ("launch-to-launch gaps of the 12-pair step (rounds 4 - 5: replayed graph; round 6: eager launches)", "DESIGN_history A")),
    ("timeline_b24.txt", ("the same at 24 pairs", "")),
    ("attn_bench.txt", ("every attention geometry of the step, fused against split / streaming kernels", "section 7, HBM-bound families")),
    ("attn_ablate.txt", ("where the fused attention kernels' time is (loads / phases switched off)", "section 7")),
    ("ln_bench.txt", ("LayerNorm forward / backward forms on cold tensors", "section 5")),
    ("gemm_fp8_cmp.txt", ("scaled-MFMA e4m3 loop against the 16x16x32 fp8 loop and bf16 on the H/14 shapes", "section 5")),
    ("tn_fp8_bench.txt", ("e4m3 weight-gradient kernel on the H/14 and B/16 shapes", "section 5")),
    ("gputest_run.log", ("`pytest -m gpu` on the final tree of the round", "")),
])
ONE_OFF = collections.OrderedDict([
    ("pmc_gemm_v6.txt", "round-1 counters of the first NT / TN kernels (superseded by pmc_mfma_util)"),
    ("pmc_gemm_nt.txt", "round-1 counters of the first 256x256 NT kernel (superseded)"),
    ("pmc_gemm_fp8_mx.txt", "wave-cycle split of the scaled-MFMA fp8 kernel against the 16x16x32 fp8 and bf16 kernels (still the e4m3 reference)"),
    ("gemm_ab_epilogue_ablation.txt", "NT epilogue ablations: no-epilogue main loop 1250-1380 TF on every shape; what stores / side loads / LDS + math cost (quoted in section 7)"),
    ("gemm_trace_tile_timeline.txt", "per-tile time stamps of every block: tile period, epilogue length, blocks inside an epilogue at a time"),
    ("gemm_kloop_ablation.txt", "K loop = MFMAs + fragment reads + LDS-DMA at 8 and 256 CUs (the 256-CU MFMA-only loop is 1.44x slower: the power limit, section 7)"),
    ("gemm_cus_scaling.txt", "per-CU rate of the NT kernel on 8 ... 256 CUs (-18 % bf16 / -27 % fp8 with all streaming)"),
    ("gemm_ab_side_prefetch.txt", "epilogue side-input prefetch variants"),
    ("gemm_ab_stagger_counted_wait_regpath.txt", "stagger / counted waits / register-path epilogue (reg* columns wrong: see regpath_fixed)"),
    ("gemm_ab_regpath_fixed.txt", "register-path epilogue, corrected: slower than the patch path"),
    ("gemm_ab_pin.txt", "pinned fragment-read order in the K loop: -1.2 % over the step's shapes (adopted)"),
    ("gemm_vs_hipblaslt.txt", "our NT / TN kernels against hipBLASLt on the step's shapes and on squares"),
    ("gemm_zero_vs_random.txt", "zero-filled operands run 15-20 % faster than random ones (the power limit seen from the GEMM side)"),
    ("gemm_w4_prototype.txt", "four-wave 128x128-wave-tile prototype: 10-30 % slower (not adopted)"),
    ("gemm_tile_choice_small_m.txt", "tile choice at small M"),
    ("gemm_two_blocks_ab.txt", "one 256x256 block per CU against two 128x128 blocks: the epilogue penalty GROWS with a partner block"),
    ("gemm_side_deriv.txt", "forward epilogue stores act'(x) instead of x (adopted, round 5)"),
    ("gate_deriv_ab.txt", "step-level A/B of the same"),
    ("gemm_ring.txt", "ring form of the 128-column NT kernel for small batches (adopted for the text tower)"),
    ("gemm_streamk_nt.txt", "stream-K walk of the NT kernel: slower at 12 / 24 pairs (opt-in)"),
    ("nt128_ring_experiment.txt", "first ring experiment (round 3)"),
    ("epilogue_side_depth.txt", "side-input prefetch depth"),
    ("cu_intake_probe.txt", "what one CU takes in from L2 / HBM by access pattern and loads in flight"),
    ("tn_ab_128_vs_256.txt", "128x128 against 256x256 weight-gradient kernel (256 adopted for the big shapes)"),
    ("tn_variants.txt", "TN kernel variants (round 3)"),
    ("tn_plan_small_m.txt", "TN split plans at small M"),
    ("tn_grouped.txt", "six weight gradients of a block in one grouped launch: window 6 000-12 000 rows"),
    ("tn_fused_reduce.txt", "in-kernel reduce of the TN partials: slower (opt-in)"),
    ("wgrad_side_stream.txt", "weight gradients on a side stream: slower (opt-in)"),
    ("adamw_ranges.txt", "AdamW range by range beside the backward: slower at every batch (opt-in)"),
    ("stream_modes_ab.txt", "hybrid residual stream against fp32 / all-bf16 streams (hybrid adopted)"),
    ("bf16_streams_ab.txt", "round-3 bf16 gradient / forward stream A/B"),
    ("attn_stagger.txt", "do the two SPACE blocks of a CU run in lockstep? no"),
    ("overlap_probe.txt", "a side-stream copy under the dgrad chain; cost of reserving 8 / 16 CUs (section 6)"),
    ("step_repro_spread_b16.txt", "the round-2 bifurcation: 36 of 300 steps on an alternative trajectory before, 0 after"),
    ("scale_run_one_gpu_box.jsonl", "tools/scale_run.sh exercised at world 1 (no node)"),
    ("bench_host_fed.txt", "round-3 host-fed / trainer rates (superseded by bench_product_path)"),
    ("bench_eager_vs_graph.jsonl", "round-3 eager against replayed step (superseded by bench_product_path: the replayed step is now slower)"),
    ("next_rows.jsonl", "validation forward + metrics (N1), downstream model (N2), uint8-fed step (N3) throughput"),
])


def main():
    files = sorted(f for f in os.listdir(P) if re.match(r"r\d\d_", f))
    by = collections.defaultdict(list)
    for f in files:
        by[f[4:]].append(f)
    out = ["# profiles/INDEX.md — which file holds the CURRENT number for each claim",
           "",
           "Generated by `tools/profiles_index.py`.  Files are `r<round>_<measurement>`; a measurement repeated in a later round supersedes the",
           "earlier files (kept for the progression tables of `DESIGN_history.md`).  `README.md` in this directory describes how each file was made.",
           "",
           "## Recurring measurements",
           "",
           "| claim | current file | quoted in DESIGN.md | superseded |",
           "|---|---|---|---|"]
    seen = set()
    for what, (claim, where) in RECURRING.items():
        fs = by.get(what, [])
        if not fs:
            continue
        seen.update(fs)
        cur, old = fs[-1], fs[:-1]
        out.append(f"| {claim} | `{cur}` | {where} | {', '.join('`' + o + '`' for o in reversed(old)) or '—'} |")
    out += ["", "## Experiments, A/Bs and negative results (one file each; valid for the kernels they name unless noted)", "",
            "| file | what it shows |", "|---|---|"]
    for what, claim in ONE_OFF.items():
        for f in by.get(what, []):
            seen.add(f)
            out.append(f"| `{f}` | {claim} |")
    rest = [f for f in files if f not in seen]
    if rest:
        out += ["", "## Other files (earlier states of a round, variants of the measurements above)", ""]
        out += [f"* `{f}`" for f in rest]
    open(os.path.join(P, "INDEX.md"), "w").write("\n".join(out) + "\n")
    print(f"{len(files)} files, {len(rest)} unclassified")


if __name__ == "__main__":
    main()
