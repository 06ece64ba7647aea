#!/usr/bin/env python3
"""Randomised bit-exactness sweep of the device-wide reduceGaussianMixture (phd_gm_reduce) against the oracle's
literal transcription of src/gm_reduce.cpp.   python tools/fuzz_gm_reduce.py [seconds=60] [first_seed=1]"""
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
    import test_gpu_eap as T
    from oracle import oracle as O
    from parity_utils import pkg
    P = pkg()
    n_ok = n_fail = 0
    t0 = time.time()
    with P.PhdFilter(P.default_config(), n_particles=4, map_capacity=64, max_measurements=8) as f:
        while time.time() - t0 < budget:
            rng = np.random.default_rng(seed)
            n = int(rng.choice([1, 2, 63, 64, 65, 129, 500, 3000, 20000, 60000]))
            centres = int(rng.choice([1, 2, 10, 64, 65, 300, max(1, n // 3)]))
            g = T.mixture(rng, n, centres, spread=float(rng.choice([0.0, 0.05, 0.3, 2.0])), extent=float(rng.choice([5, 40, 1e3])),
                          asym=bool(rng.integers(0, 2)), ties=bool(rng.integers(0, 2)))
            if rng.random() < 0.3:
                g = np.concatenate([g, g[: max(1, n // 2)]])            # exact duplicates
            if rng.random() < 0.2:
                k = rng.integers(0, len(g), 3)
                g["cov"][k[0]] = 0
                g["cov"][k[1], 0] = -1.0
                g["weight"][k[2]] = 0.0
            d = float(rng.choice([1e-6, 0.5, 10.0, 100.0, 1e6]))
            try:
                T.assert_bit_equal(f.gm_reduce(g, d), O.gm_reduce(g, d), "seed %d" % seed)
                n_ok += 1
            except AssertionError as e:
                n_fail += 1
                print("FAIL seed %d n=%d centres=%d d=%g: %s" % (seed, len(g), centres, d, str(e)[:200]))
            seed += 1
    print("gm_reduce fuzz: %d cases bit-identical, %d failed, %.0f s" % (n_ok, n_fail, time.time() - t0))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
