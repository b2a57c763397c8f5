#!/bin/bash
# HBM traffic of the MFMA GEMM launches of ONE bench step, from the TCC memory-side counters (MI355X_MICROARCH.md, "HBM"):
# two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE; they do not fit one pass), --kernel-trace only.
# usage (on the GPU box): tools/pmc_traffic.sh [bench.py flags]   -> gpurun_out/pmc_step_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch" > /dev/null 2>&1   # page the image in outside the bounded passes
STEPS=2   # --warmup 1 --steps 1, eager, no roofline / cpu legs: exactly two identical steps in the process
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o p -- \
    python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline "$@" > gpurun_out/pmc_$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - $STEPS "$@" <<'PY'
import csv, glob, json, sys, collections
steps = int(sys.argv[1]); flags = sys.argv[2:]
tot = {}
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    s = 0.0
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != c or ("gemm_nt" not in k and "gemm_tn" not in k and "tn_reduce" not in k):
            continue
        fam = "gemm_tn" if ("gemm_tn" in k or "tn_reduce" in k) else "gemm_nt"
        v = float(r["Counter_Value"]) * 1024.0          # the counters are in KiB
        per[fam][c] += v; s += v
        if c == "FETCH_SIZE": n[fam] += 1
    tot[c] = s
# gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes -> doubled (guide's correction)
sys.path.insert(0, ".")
import bench
out = {"flags": flags, "steps_in_process": steps, "gemm_source_id": bench.gemm_source_id(),
       "fetch_bytes_per_step": 2.0 * tot["FETCH_SIZE"] / steps, "write_bytes_per_step": tot["WRITE_SIZE"] / steps,
       "hbm_bytes_per_step": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps,
       "by_kernel": {k: {"fetch_bytes": 2.0 * v["FETCH_SIZE"] / steps, "write_bytes": v["WRITE_SIZE"] / steps,
                         "dispatches_per_step": n[k] / steps} for k, v in per.items()},
       "note": "FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as reported; KiB -> bytes; GEMM dispatches only "
               "(gemm_nt*, gemm_tn*, tn_reduce)"}
json.dump(out, open("gpurun_out/pmc_step_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
