"""bench.py launched the way the driver launches the scaling runs (torch.distributed.run, 2 ranks), on the one GPU of
the test box with the gloo backend: every rank must get through warm-up, the timed steps and the instrumented roofline
step (which contains the data-parallel collectives) and rank 0 must print one well-formed JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, TVTS_BENCH_ONE_DEVICE="1", TVTS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    # (259 NT launches + the weight gradients: 94 one by one, or 12 grouped + the text / sort-head ones at this small batch)
    assert d["value"] > 0 and d["roofline"]["launches"] > 280 and "cpu_baseline" not in d
    ex = d["config"]["exchange"]
    assert ex["ranks_seen"] == 2 and len(ex["devices_seen"]) == 2
    # the diagnostics a first SCALE run is read with: where the compute stream waited for a collective, per-rank step times, the
    # CU reservation in force (none unless TVTS_NT_CUS asks) and when the text tower's range was handed to the all-reduce
    for k in ("gather_wait_ms", "allreduce_exposed_ms", "allreduce_exposed_ms_max_rank", "step_ms_rank_spread"):
        assert isinstance(ex[k], float) and ex[k] >= 0.0 and ex[k] == ex[k], (k, ex[k])
    assert len(ex["step_ms_per_rank"]) == 2 and min(ex["step_ms_per_rank"]) > 0
    assert ex["cu_reservation"]["nt_cus"] is None and ex["grad_bytes_per_step"] > 0
    assert ex["text_range"].startswith("handed to the all-reduce at the join")


def _bench(extra, port):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "8",
           "--no-cpu-baseline", "--no-roofline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_graph_and_eager_bench_runs_end_at_the_same_loss():
    """bench.py launches the step eagerly (the default since round 6) or replays a captured hipGraph (--graph, world 1): the two modes
    must be the same computation -- the same loss, bit for bit, after the same number of optimizer steps (every reduction of the
    step is an ordered sum: two processes, replayed or launched, follow the same trajectory)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g = _bench(["--graph"], 0)
    e = _bench([], 0)
    assert g["config"]["hip_graph"] is True and e["config"]["hip_graph"] is False
    assert g["config"]["final_loss"] == e["config"]["final_loss"], (g["config"]["final_loss"], e["config"]["final_loss"])
    # WebVid-style batches (one caption per video, no sorting head) run through the same harness
    w = _bench(["--n-trans", "1"], 0)
    assert w["value"] > 0 and "x1" in w["config"]["workload"]
