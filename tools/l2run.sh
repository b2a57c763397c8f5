cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VARIANTS="0 0x500 0x600 0x400 0x300 0 0x500 0x600"
timeout 300 python tools/gemm_l2.py 2>&1 | grep "N="
export VARIANTS="0 0x500 0x600 0x300"
rm -rf gpurun_out/l2pmc
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/l2pmc -o p -- python tools/gemm_l2.py > gpurun_out/l2pmc.log 2>&1
python3 - <<'PY'
import csv
rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open("gpurun_out/l2pmc/p_counter_collection.csv")) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt256p" in r["Kernel_Name"]]
rows.sort()
v = [x[2] for x in rows]
per = 22
names = ["2304x768", "3072x768 act", "3072x768 gate", "768x3072", "768x768", "768x2304"]
for i in range(0, len(v), per * 4):
    print(names[i // (per * 4)], [f"{2 * 1024 * sum(v[j:j + per]) / per / 1e9:.3f} GB" for j in range(i, i + per * 4, per)])
PY
