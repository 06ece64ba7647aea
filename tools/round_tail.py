#!/usr/bin/env python3
"""Round 6, counted before building: how many unmerged survivors the merge rounds of csrc/phd_merge.h still list at the start of
each round (the rounds simulated on the oracle's survivors), i.e. from which round a ONE-SHOT finish over <= 256 (or 128) listed
survivors — merge_small's structure: all-pairs far-pair filter, exact decisions one listed pair per thread, seeds by blocks of
64, membership = first seed of the row — could replace the remaining rounds; and how many filter-positive pairs that one-shot
would have to decide exactly.
usage: python tools/round_tail.py [config = 3] [particles = 16]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
S = importlib.import_module("cuda-phdslam_amd.synthetic")
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
npart = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w = S.config_workload(cid)
cfg = O.default_config()
T = float(cfg.minSeparation)
poses = O.predict_ackerman(w["poses"], 0.05, 2.0, w["noise"][0], cfg)
rows = []
for p in np.linspace(0, w["N"] - 1, npart).astype(int):
    sv = O.update_particle(poses[p], w["maps"][p, :w["sizes"][p]], w["z"][0], cfg)["survivors"]
    order = np.lexsort((np.arange(len(sv)), -sv["weight"].astype(np.float64)))
    g = sv[order]
    n = len(g)
    m = g["mean"].astype(np.float64); c = g["cov"].astype(np.float64)
    tr = c[:, 0] + c[:, 3]
    d2 = ((m[:, None, :] - m[None, :, :]) ** 2).sum(-1)
    filt = d2 < 0.505 * T * (tr[:, None] + tr[None, :])
    s0 = 0.5 * (c[:, None, 0] + c[None, :, 0]); s1 = 0.5 * (c[:, None, 1] + c[None, :, 1]); s3 = 0.5 * (c[:, None, 3] + c[None, :, 3])
    det = s0 * s3 - s1 * s1
    dx = m[:, None, 0] - m[None, :, 0]; dy = m[:, None, 1] - m[None, :, 1]
    close = (dx * dx * s3 - 2 * dx * dy * s1 + dy * dy * s0) / det < T
    live = np.arange(n)
    hist = []
    while len(live):
        pairs = int(np.triu(filt[np.ix_(live, live)], 1).sum())
        hist.append((len(live), pairs))
        win, rest = live[:64], live[64:]
        seeds = []
        for k in win:
            if not any(close[k, s] for s in seeds):
                seeds.append(k)
        seeds = np.array(seeds)
        keep = [e for e in rest if not close[e, seeds].any()]
        live = np.array(keep, dtype=int)
    rows.append(hist)
    print("particle %5d: %4d survivors | listed at the start of each round (filter-positive pairs among them): %s" % (
        p, n, "  ".join("%d (%d)" % h for h in hist)))
nr = np.array([len(h) for h in rows])
print("\nrounds: mean %.2f" % nr.mean())
for lim in (256, 192, 128, 64):
    # rounds run before the listed count is <= lim (the one-shot replaces the rest), rounds replaced, pairs of the one-shot
    before = [next(i for i, (a, _) in enumerate(h) if a <= lim) for h in rows]
    repl = [len(h) - b for h, b in zip(rows, before)]
    at = [h[b] for h, b in zip(rows, before)]
    print("one-shot finish at <= %3d listed: after %.2f rounds on average, replaces %.2f rounds; it starts with %.0f survivors (max %d) and %.0f "
          "filter-positive pairs (max %d)" % (lim, np.mean(before), np.mean(repl), np.mean([a for a, _ in at]), max(a for a, _ in at),
                                              np.mean([q for _, q in at]), max(q for _, q in at)))
