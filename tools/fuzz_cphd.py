#!/usr/bin/env python3
"""Randomised sweep of the CPHD update (device vs oracle, tests/test_gpu_cphd.py::test_cphd_update_matches_oracle)
over random shapes, cardinality priors and max_cardinality.   python tools/fuzz_cphd.py [seconds=60] [first_seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import test_gpu_cphd as T
    n_ok = n_fail = 0
    t0 = time.time()
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        N = int(rng.integers(1, 7))
        G = int(rng.choice([1, 5, 16, 24, 48, 100]))
        M = int(rng.choice([1, 2, 10, 33, 64, 65, 70, 130]))
        nmax = int(rng.choice([7, 31, 63, 127, 255, 400]))
        try:
            T.test_cphd_update_matches_oracle(N, G, M, nmax, seed, min_structural=0.0)
            n_ok += 1
        except AssertionError as e:
            n_fail += 1
            print("FAIL seed %d N=%d G=%d M=%d nmax=%d: %s" % (seed, N, G, M, nmax, str(e)[:240].replace("\n", " ")))
        seed += 1
    print("cphd fuzz: %d cases passed, %d failed, %.0f s" % (n_ok, n_fail, time.time() - t0))
    from parity_utils import OBS
    print(OBS.report("fuzz_cphd: maxima observed, counts"))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
