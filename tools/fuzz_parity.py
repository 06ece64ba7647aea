#!/usr/bin/env python3
"""Randomised parity sweep: the per-particle update of tests/test_gpu_parity.py (merge stage bit-exact against
the oracle's merge on the device's own survivors; exact vs float moment sums; survivor sets up to members proven marginal;
the map cluster by cluster under the device's decisions with every flipped decision PROVEN from the survivor difference —
tests/parity_utils.py, compare_particle_with_oracle; particle weights) over many random shapes, seeds and configuration
corners.  Prints the maxima observed and the number of explained flips.

    python tools/fuzz_parity.py [seconds=120] [first_seed=1000]
    PHD_FUZZ_SPILL=1: dense scans of large maps on filters created with a spill list (survivor_capacity 4096): survivor lists
    beyond the LDS capacity, merged by phd_merge_spill_kernel — the same checks
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import test_gpu_parity as T
    from parity_utils import OBS, fuzz_case, pkg, synthetic
    P, S = pkg(), synthetic()
    t0 = time.time()
    n_ok = n_fail = n_skip = 0
    while time.time() - t0 < budget:
        spill = os.environ.get("PHD_FUZZ_SPILL") == "1"
        N, G, M, clustered, over = fuzz_case(seed, spill)
        cfg = P.default_config(**over)
        try:
            w = S.make_workload(N, G, M, seed=seed, clustered=clustered and G >= 8)
            cap = min(2 * G + 4 * M + 64, 1024)
            if spill:
                T.check_update_against_oracle(cfg, w, w["z"][0], cap=1024 if G > 256 else 768, mm=256, scap=4096, min_structural=0.0,
                                              structural_maps=over.get("distanceMetric", 0) == 0)
            else:
                T.check_update_against_oracle(cfg, w, w["z"][0], cap=cap, mm=max(M, 8), min_structural=0.0,
                                              structural_maps=over.get("distanceMetric", 0) == 0)
            n_ok += 1
        except P.PhdError as e:
            if e.code != -5:
                raise
            n_skip += 1                      # the random shape does not fit the configured capacities
        except Exception as e:  # noqa: BLE001
            n_fail += 1
            print("FAIL seed %d N=%d G=%d M=%d clustered=%s %s: %s" % (seed, N, G, M, clustered, over, str(e)[:300]))
            if n_fail <= 2:
                traceback.print_exc(limit=3)
        seed += 1
    print("fuzz: %d cases passed, %d failed, %d skipped (capacity), %.0f s, seeds up to %d"
          % (n_ok, n_fail, n_skip, time.time() - t0, seed - 1))
    print(OBS.report("fuzz_parity%s: maxima observed, counts" % (" (spill)" if os.environ.get("PHD_FUZZ_SPILL") == "1" else "")))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
