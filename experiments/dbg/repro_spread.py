"""Run-to-run spread of the B/16 training step (eager x3, graph x1): first-step loss / gradient norm and the three-step curve.
What tests/test_bench_path_gpu.py::test_graph_replayed_steps_match_eager_steps_and_the_oracle asserts is derived from this."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_bench_path_gpu as T  # noqa: E402
from oracle import tvts_oracle as O  # noqa: E402
from tvts_amd import _lib, arch as A  # noqa: E402

lib = _lib.load()
a, oarch = A.ARCHS["B_16"], O.ARCHS["B_16"]
P = O.synth_params(oarch, seed=21)
batch = O.synth_batch(oarch, B=4, T=8, seed=22, caption_len=32)
runs = []
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # 256: force the pipelined 256x256 NT kernel (what the test does)
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ("eager", "eager", "eager", "graph")
from tvts_amd import hip as K  # noqa: E402
K.set_default(nt_tile=TILE)
for mode in MODES:
    m, _, l, g = T._three_steps(a, P, batch, mode)
    runs.append((mode, l, g, m.store.flat.clone()))
    print(mode, ["%.9g" % x for x in l], ["%.9g" % x for x in g], flush=True)
l0, g0, f0 = runs[0][1], runs[0][2], runs[0][3]
for mode, l, g, f in runs[1:]:
    print(mode, "vs eager#0: d loss", [abs(x - y) / abs(y) for x, y in zip(l, l0)], "d gnorm", [abs(x - y) / abs(y) for x, y in zip(g, g0)],
          "params rel", T.rel(f, f0))
