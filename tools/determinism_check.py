#!/usr/bin/env python3
"""Run the 256-particle SLAM loop on the bundled simulation twice with identical inputs and report the first
step / particle at which the two runs differ bit-wise (weights, poses, map sizes, maps)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(P, data, N, steps, record):
    from e2e_utils import CLUTTER_RATE, ODOMETRY, SENSOR, VEHICLE, scan_struct
    cfg = P.default_config(**dict(SENSOR, clutterRate=CLUTTER_RATE, dt=1.0, n_particles=N, **VEHICLE, **ODOMETRY))
    rng = np.random.default_rng(2)
    out = []
    with P.PhdFilter(cfg, n_particles=N, map_capacity=512, max_measurements=64) as f:
        q = np.zeros(N, P.POSE)
        q["px"], q["py"], q["ptheta"] = data["traj"][0]
        f.set_particles(q, np.full(N, -np.log(N), np.float32))
        for k, scan in enumerate(data["scans"][:steps]):
            if k > 0:
                noise = np.stack([ODOMETRY["stdAlpha"] * rng.standard_normal(N),
                                  ODOMETRY["stdEncoder"] * rng.standard_normal(N)], 1).astype(np.float32)
                f.predict((float(data["u"][k - 1, 0]), float(data["u"][k - 1, 1])), noise)
            f.update(scan_struct(P.MEAS, scan))
            poses, lw = f.get_particles()
            maps = f.get_maps() if record else None
            did, idx = f.resample_if_needed(rng.random(), had_measurements=True)
            out.append((poses.copy(), lw.copy(), maps, did, idx.copy()))
        f.status()
    return out


def main():
    from e2e_utils import load
    P = importlib.import_module("cuda-phdslam_amd")
    data = load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 331
    a = run(P, data, N, steps, True)
    for rep in range(3):
        b = run(P, data, N, steps, True)
        for k in range(steps):
            pa, la, ma, da, ia = a[k]
            pb, lb, mb, db, ib = b[k]
            bad = []
            if not np.array_equal(pa, pb): bad.append("poses")
            if not np.array_equal(la, lb): bad.append("logw %s" % np.nonzero(la != lb)[0][:5])
            for p in range(N):
                if len(ma[p]) != len(mb[p]) or ma[p].tobytes() != mb[p].tobytes():
                    bad.append("map of particle %d (%d vs %d)" % (p, len(ma[p]), len(mb[p])))
                    if len(ma[p]) == len(mb[p]):
                        d = np.nonzero(ma[p].view(np.uint32).reshape(len(ma[p]), 7) != mb[p].view(np.uint32).reshape(len(mb[p]), 7))
                        bad.append("first differing entries %s" % (list(zip(d[0][:4], d[1][:4])),))
                    break
            if da != db or not np.array_equal(ia, ib): bad.append("resample")
            if bad:
                print("repeat %d: first difference at step %d: %s" % (rep, k, "; ".join(bad)))
                break
        else:
            print("repeat %d: identical over %d steps" % (rep, steps))


if __name__ == "__main__":
    main()
