#!/usr/bin/env python
"""GPU check of the whole-layer kernel: bf16x2 / bf16 results with and without layer fusion vs the goldens."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from parity_utils import run_fixture_on_gpu  # noqa: E402

for name in ["g0c_hd64_synth", "g1_xsmall", "g7_xsmall_refinit", "g1m_meanpool"]:
    for prec in ["bf16x2", "bf16"]:
        a = run_fixture_on_gpu(name, prec, capture=False, return_outputs=True)
        b = run_fixture_on_gpu(name, prec, capture=False, return_outputs=True, flags=32)
        d = float(np.abs(a["prune"] - b["prune"]).max())
        print(f"{name:20s} {prec:7s} fused err {a['prune_max_err']:.3e}/{a['rank_max_err']:.3e}  unfused err "
              f"{b['prune_max_err']:.3e}/{b['rank_max_err']:.3e}  |fused-unfused| {d:.3e} finite {a['finite']}", flush=True)
