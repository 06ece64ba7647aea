#!/usr/bin/env python3
"""Randomised sweep of the single-launch step (phd_step_dev: update kernel + the weights workgroup beside the merges) against
the staged calls (predict, update, resample — the unfused instantiation + the separate weights launch), bit for bit, over random
particle counts, map sizes, scan sizes, several steps each, forced and nEff-triggered resampling, PHD and CPHD.

    python tools/fuzz_fused.py [seconds=120] [first_seed=1000]
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
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import torch
    from test_gpu_parity import make_filter
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    dev = torch.device("cuda:0")
    t0 = time.time()
    n_ok = n_fail = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        N = int(rng.choice([1, 2, 7, 64, 65, 200, 256, 257, 511, 600, 1100]))
        G = int(rng.choice([1, 6, 17, 48, 96]))
        M = int(rng.choice([1, 4, 13, 32, 40]))
        if rng.random() < 0.12:
            # above 4096 particles: the block form of the weights routine as the fused tail (grid N + W) against the staged launch
            N = int(rng.choice([4097, 4608, 5000, 7000, 9001]))
            G, M = int(rng.choice([1, 6])), int(rng.choice([1, 4]))
        over = {}
        if rng.random() < 0.3:
            over.update(filterType=1, maxCardinality=int(rng.choice([31, 63, 127])))
        if rng.random() < 0.2:
            over["distanceMetric"] = 1
            over["minSeparation"] = 0.5
        cfg = P.default_config(**over)
        steps = 3
        w = S.make_workload(N, G, M, seed=seed, n_meas_sets=steps, clustered=bool(rng.integers(0, 2)) and G >= 8)
        cap = 2 * G + 4 * M + 32
        try:
            with make_filter(cfg, w, cap=cap, mm=max(M, 8)) as a, make_filter(cfg, w, cap=cap, mm=max(M, 8)) as b:
                b.debug(4)                    # never fuses
                for k in range(steps):
                    dz = torch.from_numpy(w["z"][k].view(np.uint8).copy()).to(dev)
                    dn = torch.from_numpy(w["noise"][k].copy()).to(dev)
                    torch.cuda.synchronize()
                    force = bool(rng.integers(0, 2))
                    a.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), len(w["z"][k]), w["uniform"][k], force_resample=force)
                    a.sync()
                    b.predict((2.0, 0.05), w["noise"][k])
                    b.update(w["z"][k])
                    if force:
                        b.resample(w["uniform"][k])
                    else:
                        b.resample_if_needed(w["uniform"][k], had_measurements=True)
                    pa, la = a.get_particles()
                    pb, lb = b.get_particles()
                    assert np.array_equal(pa, pb) and np.array_equal(la.view(np.uint32), lb.view(np.uint32)), "particles, step %d" % k
                    for j, (x, y) in enumerate(zip(a.get_maps(), b.get_maps())):
                        assert x.tobytes() == y.tobytes(), "map of particle %d, step %d" % (j, k)
                    if over.get("filterType"):
                        assert a.cardinalities().tobytes() == b.cardinalities().tobytes(), "cardinalities, step %d" % k
            n_ok += 1
        except P.PhdError as e:
            if e.code != -5:
                raise
        except AssertionError as e:
            n_fail += 1
            print("FAIL seed %d N=%d G=%d M=%d %s: %s" % (seed, N, G, M, over, str(e)[:200]))
        seed += 1
    print("fused-vs-staged fuzz: %d cases bit-identical, %d failed, %.0f s, seeds up to %d" % (n_ok, n_fail, time.time() - t0, seed - 1))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
