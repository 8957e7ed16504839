#!/usr/bin/env python
"""GPU box: the near-threshold decision table of tests/test_gpu_decisions.py for the three kernel sets of the row path
(> 10 000 synthetic sentences through process() on the HIP path and on the CPU oracle; thresholds swept through the
quantiles of the reference means).   python scripts/decision_flips.py > profiles/r04_decision_flips.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import test_gpu_decisions as t  # noqa: E402

print(f"{t.N_CONTEXTS} contexts x {t.SENTENCES_PER_CONTEXT} sentences, xsmall dims, synthetic O(1) weights (seed 7); 41 thresholds at the "
      f"2 % .. 98 % quantiles of the reference's sentence means; near = within {t.NEAR} of a threshold")
print(f"{'checkpoint':10s} {'kernel set':12s} {'sentences':>9s} {'near':>6s} {'flips near':>10s} {'flips > 1e-3 away':>18s} {'max |dp|':>9s} {'p99 |dp|':>9s} {'mean |dp|':>9s}")
for weights, no_f8, kernel_set in t.CASES:
    r = t.run_case(weights, no_f8, kernel_set)
    print(f"{weights:10s} {kernel_set:12s} {r['sentences']:9d} {r['near_a_threshold']:6d} {r['flips_near']:10d} {r['flips_beyond_1e-3']:18d} "
          f"{r['max_dp']:9.2e} {r['p99_dp']:9.2e} {r['mean_dp']:9.2e}", flush=True)
