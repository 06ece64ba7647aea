"""Shared helpers of the parity tests."""
import importlib

import numpy as np

from oracle import oracle as O


def pkg():
    return importlib.import_module("cuda-phdslam_amd")


def synthetic():
    return importlib.import_module("cuda-phdslam_amd.synthetic")


def oracle_config_from(cfg, **over):
    """o_config mirroring a SlamConfig (the same numbers reach both sides)"""
    oc = O.OConfig(dt=cfg.dt, minRange=cfg.minRange, maxRange=cfg.maxRange, maxBearing=cfg.maxBearing,
                   stdRange=cfg.stdRange, stdBearing=cfg.stdBearing, clutterDensity=cfg.clutterDensity, pd=cfg.pd,
                   birthWeight=cfg.birthWeight, birthNoiseFactor=cfg.birthNoiseFactor,
                   minFeatureWeight=cfg.minFeatureWeight, minSeparation=cfg.minSeparation,
                   resampleThresh=cfg.resampleThresh, l=cfg.l, h=cfg.h, a=cfg.a, b=cfg.b,
                   subdividePredict=cfg.subdividePredict, distanceMetric=cfg.distanceMetric,
                   labeledMeasurements=int(cfg.labeledMeasurements), particleWeighting=cfg.particleWeighting)
    for k, v in over.items():
        setattr(oc, k, v)
    return oc


def match_maps(got, ref):
    """Greedy nearest-neighbour matching of two Gaussian mixtures (order-insensitive).
    Returns permutation perm with got[perm[i]] matched to ref[i]."""
    assert len(got) == len(ref), (len(got), len(ref))
    n = len(ref)
    used = np.zeros(n, bool)
    perm = np.zeros(n, np.int64)
    gm, gw = got["mean"].astype(np.float64), got["weight"].astype(np.float64)
    for i in range(n):
        d = np.abs(gm - ref["mean"][i].astype(np.float64)).sum(axis=1) + np.abs(gw - float(ref["weight"][i]))
        d[used] = np.inf
        j = int(np.argmin(d))
        used[j] = True
        perm[i] = j
    return perm


def assert_maps_close(got, ref, w_rtol=5e-4, w_atol=2e-7, m_atol=2e-4, c_rtol=2e-3, c_atol=2e-6, ordered=False, what=""):
    assert len(got) == len(ref), "%s: size %d vs %d" % (what, len(got), len(ref))
    if len(ref) == 0:
        return
    g = got if ordered else got[match_maps(got, ref)]
    assert np.allclose(g["weight"], ref["weight"], rtol=w_rtol, atol=w_atol), \
        "%s: weights differ by %g" % (what, np.abs(g["weight"] - ref["weight"]).max())
    assert np.abs(g["mean"] - ref["mean"]).max() <= m_atol, \
        "%s: means differ by %g" % (what, np.abs(g["mean"] - ref["mean"]).max())
    assert np.allclose(g["cov"], ref["cov"], rtol=c_rtol, atol=c_atol), \
        "%s: covariances differ by %g" % (what, np.abs(g["cov"] - ref["cov"]).max())


def prune_margin(slab_weights, min_w):
    """min relative distance of an update-component weight to the prune threshold"""
    w = np.asarray(slab_weights, np.float64)
    if not len(w) or min_w <= 0:                # threshold 0 prunes nothing: no decision can flip
        return np.inf
    return float(np.min(np.abs(w - min_w) / min_w))


def oracle_full_update(pose, gmap, z, ocfg):
    """oracle update of one particle + the decision margins (prune, merge distance, seed weight gap)"""
    res = O.update_particle(pose, gmap, z, ocfg)
    cls = O.classify(gmap, pose, ocfg)
    f_in = gmap[cls == 1]
    pd, pre = O.preupdate(pose, f_in, z, ocfg)
    slab, flag, _ = O.update(f_in, pd, pre, O.births(pose, z, ocfg), ocfg)
    res["prune_margin"] = prune_margin(slab["weight"], ocfg.minFeatureWeight)
    res["n_in"] = int((cls == 1).sum())
    res["cls"] = cls
    return res
