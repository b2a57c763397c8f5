#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counters>" -- <cmd...>   (one rocprofv3 --pmc pass, csv output, bounded)
out=$1; ctrs=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out
timeout 90 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p -- "$@" > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "gemm" not in k and "attn" not in k and "ln_" not in k: continue
    print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
