#!/usr/bin/env python3
"""VERDICT r4 item 2, counted before anything is built: the rounds of csrc/phd_merge.h simulated on the oracle's survivors at
4096 x 256 x 64 - per round the seeds (~40 of the 64 window candidates) and, for every survivor still listed, how many of
those seeds it would have to test if the assignment culled spatially:
   dense        all the round's seeds (today)
   cells 3x3    seeds in the 3 x 3 cells around it, cell side L = sqrt(1.01 T tr_max) (tr_max over all survivors)
   x-slab       seeds with |dx| < L
   cells(2)     two size classes: cells of side L_small for the pairs of small covariances, the dense loop against the few seeds of
                large trace (tr > tr_split) - what a per-pair bound sqrt(0.505 T (tr_e + tr_s)) allows
and how many of them pass the far-pair filter / are tested exactly until the first hit.
usage: python tools/assign_cull.py [config = 3] [particles = 16]"""
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
acc = {k: 0.0 for k in ("tests", "dense", "cells", "slab", "two", "filt", "rounds", "surv", "seeds", "big_seeds", "empty_cells", "hits")}
trs = []
for p in np.linspace(0, w["N"] - 1, npart).astype(int):
    sv = O.update_particle(poses[p], w["maps"][p, :w["sizes"][p]], w["z"][0], cfg)["survivors"]
    order = np.lexsort((np.arange(len(sv)), -sv["weight"].astype(np.float64)))
    g = sv[order]
    n = len(g)
    m = g["mean"].astype(np.float64); c = g["cov"].astype(np.float64)
    tr = c[:, 0] + c[:, 3]
    trs.append(tr)
    d2 = ((m[:, None, :] - m[None, :, :]) ** 2).sum(-1)
    filt = d2 < 0.505 * T * (tr[:, None] + tr[None, :])
    s0 = 0.5 * (c[:, None, 0] + c[None, :, 0]); s1 = 0.5 * (c[:, None, 1] + c[None, :, 1]); s3 = 0.5 * (c[:, None, 3] + c[None, :, 3])
    det = s0 * s3 - s1 * s1
    dx = m[:, None, 0] - m[None, :, 0]; dy = m[:, None, 1] - m[None, :, 1]
    close = (dx * dx * s3 - 2 * dx * dy * s1 + dy * dy * s0) / det < T
    L = np.sqrt(1.01 * T * tr.max())
    x0, y0 = m[:, 0].min(), m[:, 1].min()
    cx = np.floor((m[:, 0] - x0) / L).astype(int); cy = np.floor((m[:, 1] - y0) / L).astype(int)
    tr_split = np.quantile(tr, 0.75)
    Ls = np.sqrt(1.01 * T * tr_split)
    cxs = np.floor((m[:, 0] - x0) / Ls).astype(int); cys = np.floor((m[:, 1] - y0) / Ls).astype(int)
    live = np.arange(n)
    acc["surv"] += n
    while len(live):
        win, rest = live[:64], live[64:]
        # seeds of the window: greedy inside the window
        seeds = []
        owner = {}
        for k in win:
            hit = [s for s in seeds if close[k, s]]
            if hit: owner[k] = hit[0]
            else: seeds.append(k)
        seeds = np.array(seeds)
        acc["rounds"] += 1; acc["seeds"] += len(seeds)
        big = tr[seeds] > tr_split
        acc["big_seeds"] += big.sum()
        keep = []
        for e in rest:
            acc["tests"] += 1
            acc["dense"] += len(seeds)
            near = (np.abs(cx[seeds] - cx[e]) <= 1) & (np.abs(cy[seeds] - cy[e]) <= 1)
            acc["cells"] += near.sum()
            acc["slab"] += (np.abs(m[seeds, 0] - m[e, 0]) < L).sum()
            if tr[e] > tr_split: acc["two"] += len(seeds)            # a large survivor: dense
            else:
                nearS = (np.abs(cxs[seeds] - cxs[e]) <= 1) & (np.abs(cys[seeds] - cys[e]) <= 1)
                acc["two"] += (nearS & ~big).sum() + big.sum()       # small seeds by cell, large seeds all
            f = filt[e, seeds]
            acc["filt"] += f.sum()
            hit = np.flatnonzero(f & close[e, seeds])
            if len(hit): acc["hits"] += 1
            else: keep.append(e)
        live = np.array(keep, int)
t = acc["tests"]
tr = np.concatenate(trs)
print("config %d, %d particles: %.0f survivors, %.1f rounds, %.1f seeds per round (%.1f of them with tr above the 75 %% quantile)" % (
    cid, npart, acc["surv"] / npart, acc["rounds"] / npart, acc["seeds"] / acc["rounds"], acc["big_seeds"] / acc["rounds"]))
print("  trace of the survivors' covariances: median %.4f, 75 %% %.4f, 90 %% %.4f, max %.4f m^2  ->  L(max) = %.2f m, L(75 %%) = %.2f m" % (
    np.median(tr), np.quantile(tr, 0.75), np.quantile(tr, 0.9), tr.max(), np.sqrt(1.01 * T * tr.max()), np.sqrt(1.01 * T * np.quantile(tr, 0.75))))
print("  (survivor, round) tests per particle: %.0f; seeds tested per test: dense %.1f | 3x3 cells of side L(max) %.1f | x-slab %.1f | "
      "two size classes %.1f; filter-positive %.2f; hits %.0f per particle" % (
          t / npart, acc["dense"] / t, acc["cells"] / t, acc["slab"] / t, acc["two"] / t, acc["filt"] / t, acc["hits"] / npart))
