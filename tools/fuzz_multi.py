#!/usr/bin/env python3
"""Randomised sweep of the C++ multi-device host (libphdslam_multi.so) against a single filter, bit for bit: random shard
counts (1 = a one-rank RCCL communicator, more = shards sharing this GPU through device copies), both exchange forms, host noise
or the device generator, forced / nEff-triggered resampling, a second resample without an update on some steps, steps the host
does not inspect, PHD and CPHD, degenerate and flat weight vectors.

    python tools/fuzz_multi.py [seconds=120] [first_seed=1000]
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
    from test_gpu_multi import run_single
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    MM = importlib.import_module("cuda-phdslam_amd.multi")
    t0 = time.time()
    n_ok = n_fail = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        shards = int(rng.choice([1, 2, 3, 4, 8]))
        n = int(rng.choice([1, 2, 5, 16, 40]))
        N = shards * n
        G = int(rng.choice([1, 6, 14, 40]))
        M = int(rng.choice([1, 5, 9, 16]))
        steps = 4
        exchange = str(rng.choice(["gathered", "alltoall", "pull"]))
        device_rng = bool(rng.integers(0, 2))
        over = dict(n_particles=N, resampleThresh=float(rng.choice([0.3, 0.6, 0.9])))
        if rng.random() < 0.3:
            over.update(filterType=1, maxCardinality=63)
        cfg = P.default_config(**over)
        w = S.make_workload(N, G, M, seed=seed, n_meas_sets=steps)
        w["logw"] = (w["logw"] + rng.choice([0.0, 3.0, 12.0]) * np.linspace(0, 1.0, N).astype(np.float32)).astype(np.float32)
        force = [bool(x) for x in rng.integers(0, 2, steps)]
        # a second resample with no update in between on some steps (the shards' copy-free resample falls back to the copying one
        # there: its indirection still names guest slabs), and steps after which the host does not look at the maps
        extra = [bool(x) for x in (rng.random(steps) < 0.25)]
        look = [bool(x) for x in (rng.random(steps) < 0.7)]
        look[-1] = True
        cap, mm = 2 * G + 4 * M + 32, 16
        try:
            ref = run_single(cfg, w, steps, cap, mm, device_rng, force, extra)
            with MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=cap, max_measurements=mm,
                                exchange={"gathered": MM.EXCHANGE_GATHERED, "alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL}[exchange]) as m:
                m.seed(77)
                m.set_particles(w["poses"], w["logw"])
                m.set_maps(w["maps"], w["sizes"])
                for k in range(steps):
                    did = m.step((2.0, 0.05 - 0.01 * k), None if device_rng else w["noise"][k], w["z"][k], w["uniform"][k],
                                 force_resample=force[k])
                    if extra[k]:
                        m.resample(0.5 * float(w["uniform"][k]))
                    if not look[k]:
                        continue
                    p, lw = m.get_particles()
                    rdid, rp, rlw, rmaps = ref[k]
                    assert did == rdid, "resample decision, step %d" % k   # (of the step itself; the extra resample is unconditional)
                    assert np.array_equal(p, rp) and np.array_equal(lw.view(np.uint32), rlw.view(np.uint32)), "particles, step %d" % k
                    for j, (a, b) in enumerate(zip(m.get_maps(), rmaps)):
                        assert a.tobytes() == b.tobytes(), "map of particle %d, step %d" % (j, k)
            n_ok += 1
        except AssertionError as e:
            n_fail += 1
            print("FAIL seed %d shards=%d n=%d G=%d M=%d %s rng=%s %s: %s" % (seed, shards, n, G, M, exchange, device_rng, over, str(e)[:200]))
        seed += 1
    print("multi-host fuzz: %d cases bit-identical to a single filter, %d failed, %.0f s, seeds up to %d" % (n_ok, n_fail, time.time() - t0, seed - 1))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
