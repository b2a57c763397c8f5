"""Data-parallel path of the HIP step engine with two ranks sharing the one GPU of the test box (gloo backend moving
CUDA tensors: the collectives' semantics are the backend-independent part; RCCL itself needs >1 GPU).  Checked against
the oracle's multi-rank step: all-gather of embeddings (AllGather_multi, trainer.py:41-57), local-row backward,
range-by-range gradient all-reduce with the 1/W average."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["b16_style", "h14"])
def test_two_ranks_match_oracle(variant):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "dist_smoke.py")] + (["h14"] if variant == "h14" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "DIST_SMOKE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
