#!/usr/bin/env python
"""Precision-policy sweep on the GPU: max |logit error| against the reference goldens per policy per fixture.

Every policy is evaluated through the C ABI (`OP_PRECISION_CUSTOM`); policies without a curated kernel set run on
the all-terms kernels with the unused lo operands cleared -- bit-identical numerics to a kernel that omits the term
(tests/test_gpu_policy.py checks that identity), so the table is about arithmetic only, not speed.

    python scripts/precision_sweep.py [--out gpurun_out/precision_sweep.json] [--fixtures g1_xsmall,...]
"""

from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from open_provence_amd._lib import OP_FAMILIES  # noqa: E402

DEFAULT_FIXTURES = ["g0b_hd64_refinit", "g0c_hd64_synth", "g1_xsmall", "g1m_meanpool", "g2_gte_varlen"]


def policies() -> dict[str, dict[str, int]]:
    full = {f: 3 for f in OP_FAMILIES}
    out: dict[str, dict[str, int]] = {"bf16x3 (all terms)": dict(full)}
    # one family at a time: drop lo(weight/right) (mask 1), drop lo(activation/left) (mask 2), drop both (mask 0)
    for fam in OP_FAMILIES:
        for mask in (1, 2, 0):
            p = dict(full)
            p[fam] = mask
            out[f"{fam}={mask}"] = p
    gemms = ("wqkv", "attn_out", "wi", "mlp_out")
    out["weights bf16 (4 GEMMs = 1)"] = {**full, **{f: 1 for f in gemms}}
    out["activations bf16 in GEMMs (4 GEMMs = 2)"] = {**full, **{f: 2 for f in gemms}}
    out["4 GEMMs = 0"] = {**full, **{f: 0 for f in gemms}}
    out["attention = 0 (qk, pv)"] = {**full, "qk": 0, "pv": 0}
    out["pv = 2, qk = 3 (no lo(p))"] = {**full, "pv": 2}
    out["weights bf16 + pv = 2"] = {**full, **{f: 1 for f in gemms}, "pv": 2}
    out["bf16 (single pass)"] = {f: 0 for f in OP_FAMILIES}
    return out


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--out", default=str(ROOT / "gpurun_out" / "precision_sweep.json"))
    parser.add_argument("--fixtures", default=",".join(DEFAULT_FIXTURES))
    args = parser.parse_args()

    from parity_utils import run_fixture_on_gpu

    rows = []
    for name in args.fixtures.split(","):
        for label, pol in policies().items():
            rep = run_fixture_on_gpu(name, pol, capture=False)
            rows.append({"fixture": name, "policy": label, "terms": pol, "prune": rep["prune_max_err"], "rank": rep["rank_max_err"],
                         "keep_prob": rep["keep_prob_max_err"], "kernel_set": rep.get("kernel_set")})
            print(f"{name:20s} {label:42s} prune {rep['prune_max_err']:.2e} rank {rep['rank_max_err']:.2e} "
                  f"keep {rep['keep_prob_max_err']:.2e} [{rep.get('kernel_set')}]", flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
