#!/usr/bin/env python3
"""What a sparse-neighbour greedy merge would have to do at 4096 x 256 x 64, counted on the CPU from the
oracle's survivors: per particle the number of survivors, of (later, earlier) pairs the conservative far-pair filter passes,
of exactly close pairs, the candidates a 3 x 3 grid neighbourhood of cell side L = sqrt(1.01 T tr_max) holds, the number of
parallel sweeps the fixed point seed[i] = not any(close(i, j) and seed[j], j earlier) needs — against what the rounds of
csrc/phd_merge.h do today (PHD_ASSIGN_STATS: 1 124 (survivor, round) tests x ~40 seeds, 955 + ~320 exact decisions).
usage: python tools/merge_structure.py [config = 3] [particles = 24]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
S = importlib.import_module("cuda-phdslam_amd.synthetic")
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
npart = int(sys.argv[2]) if len(sys.argv) > 2 else 24
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
    filt = d2 < 0.505 * T * (tr[:, None] + tr[None, :])                      # the kernel's far-pair filter
    s0 = 0.5 * (c[:, None, 0] + c[None, :, 0]); s1 = 0.5 * (c[:, None, 1] + c[None, :, 1]); s3 = 0.5 * (c[:, None, 3] + c[None, :, 3])
    det = s0 * s3 - s1 * s1
    dx = m[:, None, 0] - m[None, :, 0]; dy = m[:, None, 1] - m[None, :, 1]
    d = (dx * dx * s3 - 2 * dx * dy * s1 + dy * dy * s0) / det
    close = d < T
    low = np.tril(np.ones((n, n), bool), -1)                                  # (i later, j earlier)
    nf, nc = int((filt & low).sum()), int((close & low).sum())
    # fixed point by parallel sweeps
    adj = close & low
    seed = np.ones(n, bool); sweeps = 0
    while True:
        new = ~(adj & seed[None, :]).any(1)
        sweeps += 1
        if np.array_equal(new, seed): break
        seed = new
    # greedy (reference) for the cluster count and a cross-check of the fixed point
    merged = np.zeros(n, bool); k = 0; gseed = np.zeros(n, bool)
    for i in range(n):
        if merged[i]: continue
        gseed[i] = True; k += 1
        merged |= close[i] & ~merged
    assert np.array_equal(gseed, seed), "fixed point != greedy"
    # 3 x 3 grid neighbourhood
    L = np.sqrt(1.01 * T * tr.max())
    cx = np.floor((m[:, 0] - m[:, 0].min()) / L).astype(int); cy = np.floor((m[:, 1] - m[:, 1].min()) / L).astype(int)
    near = (np.abs(cx[:, None] - cx[None, :]) <= 1) & (np.abs(cy[:, None] - cy[None, :]) <= 1)
    rows.append((n, k, nf, nc, int((near & low).sum()), sweeps, L, int(adj.sum(1).max())))
r = np.array(rows, float)
print("config %d, %d particles: survivors %.0f (max %.0f), clusters %.0f" % (cid, npart, r[:, 0].mean(), r[:, 0].max(), r[:, 1].mean()))
print("  (later, earlier) pairs the far-pair filter passes: %.0f per particle (%.1f per survivor); exactly close: %.0f (%.1f per "
      "survivor, at most %.0f for one survivor)" % (r[:, 2].mean(), (r[:, 2] / r[:, 0]).mean(), r[:, 3].mean(), (r[:, 3] / r[:, 0]).mean(), r[:, 7].max()))
print("  3 x 3 cells of side L = %.2f m (sqrt(1.01 T tr_max)): %.0f earlier candidates per particle (%.1f per survivor)" %
      (r[:, 6].mean(), r[:, 4].mean(), (r[:, 4] / r[:, 0]).mean()))
print("  parallel sweeps of the seed fixed point: mean %.1f, max %.0f" % (r[:, 5].mean(), r[:, 5].max()))
print("  the rounds today (profiles/r03_phase_stamps.txt, PHD_ASSIGN_STATS): ~45 000 filter tests in the assignment + ~16 000 in the "
      "window matrices, ~955 + ~320 exact decisions, 4.1 rounds x 3 barriers")
