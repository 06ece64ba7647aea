#!/usr/bin/env python3
"""Randomised parity sweep: the per-particle update of tests/test_gpu_parity.py (merge stage bit-exact against
the oracle's merge on the device's own survivors, update stage against the oracle where no prune/merge decision
is fp-marginal, particle weights) over many random shapes, seeds and configuration corners.

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
    from parity_utils import pkg, synthetic
    P, S = pkg(), synthetic()
    t0 = time.time()
    n_ok = n_fail = n_skip = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        N = int(rng.integers(1, 9))
        G = int(rng.choice([1, 3, 17, 32, 64, 100, 160, 256]))
        M = int(rng.choice([1, 2, 7, 16, 32, 33, 64, 65, 128]))
        clustered = bool(rng.integers(0, 2))
        over = {}
        if rng.random() < 0.25:
            over["distanceMetric"] = 1
            over["minSeparation"] = float(rng.choice([0.2, 0.5, 0.8]))
        elif rng.random() < 0.3:
            over["minSeparation"] = float(rng.choice([0.5, 3.0, 10.0, 40.0]))
        if rng.random() < 0.2:
            over["minFeatureWeight"] = float(rng.choice([1e-8, 1e-4, 1e-2]))
        if rng.random() < 0.2:
            over["maxRange"] = float(rng.choice([6.0, 10.0]))
        if rng.random() < 0.15:
            over["birthWeight"] = float(rng.choice([1e-3, 0.05]))
        spill = os.environ.get("PHD_FUZZ_SPILL") == "1"
        if spill:
            N = int(rng.integers(1, 4))
            G = int(rng.choice([160, 256, 320]))
            M = int(rng.choice([128, 200, 256]))
            clustered = True
            over = {k: v for k, v in over.items() if k in ("distanceMetric", "minSeparation")}
            if rng.random() < 0.3:
                over["clutterRate"] = float(rng.choice([50.0, 150.0]))
        cfg = P.default_config(**over)
        try:
            w = S.make_workload(N, G, M, seed=seed, clustered=clustered and G >= 8)
            cap = min(2 * G + 4 * M + 64, 1024)
            if spill:
                T.check_update_against_oracle(cfg, w, w["z"][0], cap=1024 if G > 256 else 768, mm=256, scap=4096, min_structural=0.0,
                                              structural_maps=False)
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
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
