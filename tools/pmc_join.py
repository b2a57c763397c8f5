"""Join the launch-ordered GEMM list of one bench step (TVTS_BENCH_ORDER=… bench.py) with the per-dispatch PMC rows of
tools/pmc_traffic.sh: measured HBM bytes (FETCH_SIZE x2 + WRITE_SIZE) against the algorithmic bytes, per GEMM shape."""
import collections
import csv
import json
import sys

order = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_order.json"))
root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"


def rows(c):
    """per-kind value lists of the SECOND of the two identical steps; a tn_reduce dispatch is added to the gemm_tn dispatch it
    follows (weight gradients whose contraction is not split have none)"""
    ev = []
    for r in csv.DictReader(open(f"{root}/pmc_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        v = float(r["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1)
        kind = "red" if "tn_reduce" in k else "gemm_tn" if "gemm_tn" in k else "gemm_nt" if "gemm_nt" in k else None
        if kind:
            ev.append((int(r["Dispatch_Id"]), kind, v))
    ev.sort()
    out = {"gemm_nt": [], "gemm_tn": []}
    for _, kind, v in ev:
        if kind == "red":
            out["gemm_tn"][-1] += v
        else:
            out[kind].append(v)
    for k in out:
        out[k] = out[k][len(out[k]) // 2:]          # two identical steps in the process: keep the second
    return out, None


F, Fr = rows("FETCH_SIZE")
W, Wr = rows("WRITE_SIZE")
idx = {"gemm_nt": 0, "gemm_tn": 0}
agg = collections.OrderedDict()
for kind, shp, ms, alg in order:
    i = idx[kind]; idx[kind] += 1
    f, w = F[kind][i], W[kind][i]
    d = agg.setdefault((kind,) + tuple(shp), [0, 0.0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += ms; d[2] += f; d[3] += w; d[4] += alg
assert idx["gemm_nt"] == len(F["gemm_nt"]) and idx["gemm_tn"] == len(F["gemm_tn"]), (idx, len(F["gemm_nt"]), len(F["gemm_tn"]))
print(f"{'ms':>7} {'n':>3} {'fetch GB':>9} {'write GB':>9} {'alg GB':>8} {'ratio':>6} {'TB/s':>5}  shape")
T = [0, 0, 0, 0]
for k, (n, ms, f, w, alg) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:7.2f} {n:3d} {f / 1e9:9.2f} {w / 1e9:9.2f} {alg / 1e9:8.2f} {(f + w) / alg:6.2f} {(f + w) / ms / 1e9:5.2f}  {k}")
    T[0] += ms; T[1] += f; T[2] += w; T[3] += alg
print(f"{T[0]:7.2f}     {T[1] / 1e9:9.2f} {T[2] / 1e9:9.2f} {T[3] / 1e9:8.2f} {(T[1] + T[2]) / T[3]:6.2f} {(T[1] + T[2]) / T[0] / 1e9:5.2f}  total")
