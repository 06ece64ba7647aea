#!/usr/bin/env python3
"""Randomised sweep of the weights routine (normalise, nEff, fixed-point CDF, systematic / stratified indices) against the
oracle: indices bit for bit, over random particle counts (every instantiation: 256 / 512 / 1024 / 4096 / 16384 thresholds, the
split kernel above 1024, the chunked kernel above 16384), weight spreads from flat to one particle carrying everything,
un-normalised vectors (overflow guard), ties and -inf-like weights.

    python tools/fuzz_resample.py [seconds=60] [first_seed=1000]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    from oracle import oracle as O
    P = importlib.import_module("cuda-phdslam_amd")
    t0 = time.time()
    n_ok = n_fail = 0
    cfg = P.default_config()
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 2, 3, 63, 64, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2048, 3000, 4095, 4096, 4097,
                            9999, 16384, 16385, 40000]))
        if rng.random() < 0.5:
            n = int(rng.integers(1, 20001))
        sigma = float(rng.choice([0.0, 0.1, 2.0, 10.0, 40.0]))
        lw = rng.normal(0, sigma, n).astype(np.float32)
        if rng.random() < 0.2:
            lw[rng.integers(0, n, max(1, n // 10))] = np.float32(-1e30)      # dead particles
        if rng.random() < 0.2:
            lw[:] = lw[rng.integers(0, n)]                                  # all equal
        if rng.random() < 0.7:
            lw = O.normalize_weights(lw)
        else:
            lw = (O.normalize_weights(lw) + np.float32(rng.choice([-0.7, 0.3]))).astype(np.float32)   # sums to != 1
        poses = np.zeros(n, P.POSE)
        poses["px"] = np.arange(n)
        try:
            with P.PhdFilter(cfg, n_particles=n, map_capacity=8, max_measurements=8) as f:
                f.set_particles(poses, lw)
                ref_neff = O.neff(lw)
                if np.isfinite(ref_neff):
                    # the oracle adds the n terms one after the other in float like the reference (src/main.cpp:1281-1284):
                    # up to n/2 ulps of accumulated rounding; the device's tree is the more accurate of the two
                    assert abs(f.neff() - ref_neff) <= (2e-5 + 6e-8 * n) * max(1.0, abs(ref_neff)), ("neff", f.neff(), ref_neff)
                if rng.random() < 0.7:
                    u = float(rng.uniform())
                    idx = f.resample(u)
                    assert np.array_equal(idx, O.resample(lw, u)), "systematic indices"
                else:
                    us = rng.uniform(0, 1, n)
                    idx = f.resample(us)
                    assert np.array_equal(idx, O.resample(lw, us)), "stratified indices"
                p2, lw2 = f.get_particles()
                assert np.array_equal(p2["px"], poses["px"][idx]), "copy_particles"
                assert np.all(lw2 == np.float32(-np.log(float(n)))), "weights after the resample"
            n_ok += 1
        except AssertionError as e:
            n_fail += 1
            print("FAIL seed %d n=%d sigma=%g: %s" % (seed, n, sigma, str(e)[:200]))
        seed += 1
    print("resample fuzz: %d cases passed (indices bit-exact), %d failed, %.0f s, seeds up to %d" % (n_ok, n_fail, time.time() - t0, seed - 1))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
