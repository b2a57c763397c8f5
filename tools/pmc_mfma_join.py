"""Join the two PMC passes of tools/pmc_mfma_util.sh (SQ counters; GRBM clock counters) with the launch-ordered GEMM list of the
same step: matrix-pipe duty, wait / issue-stall shares and the clock under load, per GEMM shape and per attention / LayerNorm kernel.

  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE)   (rocprofv3's MfmaUtil formula: busy cycles summed over the SIMDs)
  clock      = GRBM_GUI_ACTIVE / traced duration of the same dispatch (MI355X_MICROARCH.md, "DVFS give-back")
  wait / stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of the wave-cycles)
  peak@clock = 256 CUs x 4 SIMDs x 1024 FLOP per cycle x clock  (v_mfma_f32_16x16x32_bf16: 16 384 FLOP in 16 cycles per SIMD)
usage: pmc_mfma_join.py gemm_order.json <root with pmcu_sq/ pmcu_grbm/> [bench flags ...]  -> table on stdout, <root>/pmc_mfma_util.json"""
import collections
import csv
import glob
import json
import os
import sys

order = json.load(open(sys.argv[1]))
root = sys.argv[2]
flags = sys.argv[3:]
SIMDS = 1024


def load(p):
    """dispatch id -> {name, counters, dur_ns}"""
    d = {}
    f = glob.glob(f"{root}/pmcu_{p}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": collections.defaultdict(float), "dur": None})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            e["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    kt = glob.glob(f"{root}/pmcu_{p}/**/*kernel_trace.csv", recursive=True)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            i = int(r["Dispatch_Id"])
            if i in d:
                d[i]["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return d


def kind_of(name):
    if "tn_reduce" in name:
        return "red"
    if "gemm_tn" in name:
        return "gemm_tn"
    if "gemm_nt" in name:
        return "gemm_nt"
    return None


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("tvts::", "")
    return n[:70]


def keyed(d):
    """second of the two identical steps: GEMM dispatches keyed by (kind, shape) through the launch order, the rest by kernel name"""
    ids = sorted(d)
    seq = {"gemm_nt": [], "gemm_tn": []}
    other = collections.defaultdict(list)
    for i in ids:
        k = kind_of(d[i]["name"])
        if k in seq:
            seq[k].append(i)
        elif k is None:
            other[short(d[i]["name"])].append(i)
    out = collections.OrderedDict()
    idx = {"gemm_nt": 0, "gemm_tn": 0}
    half = {k: len(v) // 2 for k, v in seq.items()}
    n_order = collections.Counter(k for k, *_ in order)
    for k in seq:
        assert len(seq[k]) - half[k] == n_order[k], (k, len(seq[k]), half[k], n_order[k])
    for kind, shp, ms, alg in order:
        i = seq[kind][half[kind] + idx[kind]]
        idx[kind] += 1
        out.setdefault((kind,) + tuple(shp), []).append(i)
    for n, v in other.items():
        if any(t in n for t in ("attn", "ln_", "adamw")):
            out[("other", n)] = v[len(v) // 2:]
    return out


sq, gr = load("sq"), load("grbm")
ks, kg = keyed(sq), keyed(gr)
rows = []
for key in ks:
    a = [sq[i] for i in ks[key]]
    b = [gr[i] for i in kg.get(key, [])]
    n = len(a)
    c = collections.defaultdict(float)
    for e in a:
        for k, v in e["c"].items():
            c[k] += v
    dur_sq = sum(e["dur"] or 0.0 for e in a)
    gui = sum(e["c"]["GRBM_GUI_ACTIVE"] for e in b)
    dur_gr = sum(e["dur"] or 0.0 for e in b)
    if key[0] != "other":
        M, N, K = key[1], key[2], key[3]
        flop = 2.0 * M * N * K * n
    else:
        flop = 0.0
    clock = gui / dur_gr if dur_gr else None     # cycles per ns = GHz
    if clock and clock > 4.0:                      # the counter summed over the 8 XCDs
        clock /= 8.0
        gui /= 8.0
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    # kernel cycles of the SQ pass, from its own traced duration at the clock of the GRBM pass
    cyc_sq = (dur_sq * clock) if clock else None
    rows.append({
        "key": list(key), "n": n, "us_sq_pass": dur_sq / n / 1e3, "us_grbm_pass": dur_gr / max(len(b), 1) / 1e3,
        "tflops_sq_pass": flop / dur_sq / 1e3 if flop and dur_sq else None,
        "clock_ghz": clock,
        "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc_sq) if cyc_sq else None,
        "mfma_cycles_per_flop_check": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (flop / 16384.0)) if flop else None,
        "mops_bf16": c["SQ_INSTS_VALU_MFMA_MOPS_BF16"],
        "wait_any": c["SQ_WAIT_ANY"] / wc, "wait_inst_any": c["SQ_WAIT_INST_ANY"] / wc, "active_inst_any": c["SQ_ACTIVE_INST_ANY"] / wc,
        "wait_inst_lds": c["SQ_WAIT_INST_LDS"] / wc, "sq_busy_cycles": c["SQ_BUSY_CYCLES"], "wave_cycles": wc,
        "peak_tflops_at_clock": 1024 * 1024 * clock / 1e3 if clock else None, "total_ms": dur_sq / 1e6, "flop": flop,
    })
rows.sort(key=lambda r: -r["total_ms"])


def fmt(x, f):
    return format(x, f) if x is not None else "   -"


print(f"# tools/pmc_mfma_util.sh {' '.join(flags)}: one eager bench step, per-dispatch counters of the SECOND of two identical steps; "
      f"profiled dispatches run serialised")
print(f"{'ms':>7} {'n':>3} {'us':>7} {'TF':>6} {'GHz':>5} {'peak@clk':>8} {'TF/peak':>7} {'mfma_busy':>9} {'cyc/mfma':>8} {'wait':>5} {'stall':>5} {'active':>6} {'ldsst':>5}  kernel")
for r in rows:
    fr = (r["tflops_sq_pass"] / r["peak_tflops_at_clock"]) if r["tflops_sq_pass"] and r["peak_tflops_at_clock"] else None
    print(f"{r['total_ms']:7.2f} {r['n']:3d} {r['us_sq_pass']:7.1f} {fmt(r['tflops_sq_pass'], '6.0f')} {fmt(r['clock_ghz'], '5.2f')} "
          f"{fmt(r['peak_tflops_at_clock'], '8.0f')} {fmt(fr, '7.3f')} {fmt(r['mfma_busy'], '9.3f')} {fmt(r['mfma_cycles_per_flop_check'], '8.1f')} "
          f"{r['wait_any']:5.2f} {r['wait_inst_any']:5.2f} {r['active_inst_any']:6.2f} {r['wait_inst_lds']:5.2f}  {tuple(r['key'])}")
# family totals over the GEMM launches
tot = {}
for fam in ("gemm_nt", "gemm_tn"):
    rs = [r for r in rows if r["key"][0] == fam and r["clock_ghz"]]
    if not rs:
        continue
    ms = sum(r["total_ms"] for r in rs)
    fl = sum(r["flop"] for r in rs)
    clk = sum(r["clock_ghz"] * r["total_ms"] for r in rs) / ms
    busy = sum((r["mfma_busy"] or 0.0) * r["total_ms"] for r in rs) / ms
    tot[fam] = {"ms": ms, "tflops": fl / ms / 1e9, "clock_ghz": clk, "mfma_busy": busy, "peak_tflops_at_clock": 1024 * 1024 * clk / 1e3}
rs = [r for r in rows if r["key"][0] in ("gemm_nt", "gemm_tn") and r["clock_ghz"]]
ms = sum(r["total_ms"] for r in rs)
allg = {"ms": ms, "tflops": sum(r["flop"] for r in rs) / ms / 1e9,
        "clock_ghz": sum(r["clock_ghz"] * r["total_ms"] for r in rs) / ms,
        "mfma_busy": sum((r["mfma_busy"] or 0.0) * r["total_ms"] for r in rs) / ms}
allg["peak_tflops_at_clock"] = 1024 * 1024 * allg["clock_ghz"] / 1e3
print(f"# all MFMA GEMM launches: {allg['ms']:.2f} ms, {allg['tflops']:.0f} TF in the profiled pass, clock under load {allg['clock_ghz'] * 1e3:.0f} MHz "
      f"-> dense bf16 peak at that clock {allg['peak_tflops_at_clock']:.0f} TF (sheet: 2500 at 2400 MHz), matrix-pipe duty {allg['mfma_busy']:.3f}")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
json.dump({"flags": flags, "gemm_source_id": bench.gemm_source_id(), "all_gemm": allg, "by_family": tot, "rows": rows,
           "note": "rocprofv3 --pmc, two passes (SQ; GRBM) + --kernel-trace; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel "
                   "cycles), clock = GRBM_GUI_ACTIVE / traced duration; profiled dispatches are serialised and run at the profiled clock"},
          open(f"{root}/pmc_mfma_util.json", "w"), indent=1)
