#!/usr/bin/env python3
"""Time of the weights / nEff / resample routine as a launch of its own (HIP events around the launch, phd_timing_*), by particle
count: the one-workgroup forms up to 4096, the block form (round 5: several workgroups, two grid-wide barriers) above.
usage: python3 tools/weights_time.py [n ...]"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
P = importlib.import_module("cuda-phdslam_amd")
from oracle import oracle as O   # noqa: E402  (the checker: indices against the oracle's)

ns = [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192, 16384, 65536]
for n in ns:
    rng = np.random.default_rng(n)
    lw = O.normalize_weights((rng.normal(0, 2.0, n) + np.linspace(0, 6, n)).astype(np.float32))
    with P.PhdFilter(P.default_config(), n_particles=n, map_capacity=8, max_measurements=8) as f:
        f.set_particles(None, lw)
        f.set_frozen(True)
        for mode in ("resample (indices + copy_particles from normalised weights)",):
            for _ in range(20):
                f.resample(0.37)
            f.timing_reset(); f.timing(True)
            for _ in range(200):
                idx = f.resample(0.37)
            f.timing(False)
            ms, k = f.timing_read()
            ok = np.array_equal(idx, O.resample(lw, 0.37))
            print("n = %6d  %-64s %7.2f us per launch (%d launches)  indices == oracle: %s" % (
                n, mode, 1e3 * ms[P._lib.K_WEIGHTS] / max(k[P._lib.K_WEIGHTS], 1), k[P._lib.K_WEIGHTS], ok))
