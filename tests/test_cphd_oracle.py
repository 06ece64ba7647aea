"""The CPHD oracle (oracle/cphd_cpu.c, parity unpinned — see its header) against two independent
statements of the same recursion:

 * an arbitrary-precision (decimal, 60 digits) evaluation of Vo, Vo & Cantoni's GM-CPHD update
   (IEEE TSP 2007, eqs. 31-35 / 44-49) written directly in the linear domain, with the elementary
   symmetric functions taken by SUBSET ENUMERATION for small sets and by polynomial expansion for
   larger ones — no log-domain recursion, no shared code with the oracle;
 * HEAD's PHD update (oracle o_update_particle): with a Poisson cardinality prior the CPHD
   recursion must reduce to it.
"""
import itertools
import math
from decimal import Decimal as Dm, getcontext

import numpy as np
import pytest

from oracle import oracle as O

getcontext().prec = 60


def esf_enum(xs):
    e = [Dm(0)] * (len(xs) + 1)
    for j in range(len(xs) + 1):
        for sub in itertools.combinations(xs, j):
            pr = Dm(1)
            for x in sub:
                pr *= x
            e[j] += pr
    return e


def esf_poly(xs):
    e = [Dm(1)]
    for x in xs:
        e = [(e[j] if j < len(e) else Dm(0)) + (x * e[j - 1] if j >= 1 else Dm(0)) for j in range(len(e) + 1)]
    return e


def brute(cn_prior, S, w_all, pdw, bw, lam, kap, esf=esf_poly):
    M, Nmax = len(S), len(cn_prior) - 1
    bw, lam, kap = Dm(float(np.float32(bw))), Dm(float(np.float32(lam))), Dm(float(np.float32(kap)))
    p0 = [Dm(float(x)).exp() for x in cn_prior]
    binom = [Dm(math.comb(M, k)) * bw ** k * (1 - bw) ** (M - k) for k in range(M + 1)]
    pp = [sum(binom[k] * p0[n - k] for k in range(min(n, M) + 1)) for n in range(Nmax + 1)]
    xi = [(lam / kap) * (Dm(float(s)) + bw) for s in S]
    W1 = Dm(float(w_all)) + M * bw
    Wq = Dm(float(w_all)) - Dm(float(pdw))

    def upsilon(u, xs):
        ms = len(xs)
        e = esf(xs)
        out = []
        for n in range(Nmax + 1):
            acc = Dm(0)
            for j in range(min(ms, n) + 1):
                if n >= j + u:
                    acc += lam ** (ms - j) * (-lam).exp() * Dm(math.perm(n, j + u)) * Wq ** (n - j - u) / W1 ** n * e[j]
            out.append(acc)
        return out

    Y0, Y1 = upsilon(0, xi), upsilon(1, xi)
    y0 = sum(a * b for a, b in zip(Y0, pp))
    y1 = sum(a * b for a, b in zip(Y1, pp))
    lz = []
    for m in range(M):
        d = sum(a * b for a, b in zip(upsilon(1, xi[:m] + xi[m + 1:]), pp))
        lz.append(float(-((lam / kap) * d / y0).ln()))
    cn = [float((pp[n] * Y0[n] / y0).ln()) if pp[n] * Y0[n] > 0 else -np.inf for n in range(Nmax + 1)]
    return dict(lz=np.array(lz), r1=float(y1 / y0), cn=np.array(cn), lY0=float(y0.ln()))


def log_poisson(mean, nmax):
    n = np.arange(nmax + 1)
    return (n * np.log(mean) - mean - np.array([math.lgamma(k + 1) for k in n])).astype(np.float32)


def random_case(rng, M, nmax, n_targets):
    """S_m of a plausible scene: a few well-detected targets (large S), the rest clutter-like"""
    S = rng.uniform(0, 0.02, M)
    det = rng.choice(M, min(n_targets, M), replace=False)
    S[det] = rng.uniform(5.0, 60.0, len(det))
    w_all = float(n_targets * rng.uniform(0.8, 1.1))
    pdw = 0.95 * w_all * rng.uniform(0.5, 1.0)
    prior = rng.normal(0, 1.5, nmax + 1) - 0.15 * (np.arange(nmax + 1) - n_targets) ** 2
    prior = (prior - np.log(np.exp(prior).sum())).astype(np.float32)
    return prior, S.astype(np.float32), w_all, pdw


BW, LAM, KAP = 1e-4, 20.0, 20.0 / (2 * np.pi * 15.0)


@pytest.mark.parametrize("M,nmax,nt,seed", [(1, 6, 1, 0), (3, 10, 2, 1), (6, 12, 3, 2), (7, 20, 9, 3), (5, 4, 2, 4)])
def test_terms_against_subset_enumeration(M, nmax, nt, seed):
    rng = np.random.default_rng(seed)
    prior, S, w_all, pdw = random_case(rng, M, nmax, nt)
    got = O.cphd_terms(prior, S, w_all, pdw, BW, LAM, KAP)
    ref = brute(prior, S, w_all, pdw, BW, LAM, KAP, esf=esf_enum)
    assert np.allclose(got["lz"], ref["lz"], atol=2e-4)
    assert abs(got["r1"] - ref["r1"]) < 2e-4 * max(1.0, ref["r1"])
    assert abs(got["lY0"] - ref["lY0"]) < 2e-4 * max(1.0, abs(ref["lY0"]))
    live = ref["cn"] > -60
    assert np.allclose(got["cn"][live], ref["cn"][live], atol=5e-4)
    assert abs(np.log(np.exp(got["cn"].astype(np.float64)).sum())) < 1e-4        # the posterior is normalised


@pytest.mark.parametrize("M,nmax,nt,seed", [(32, 63, 12, 10), (64, 255, 40, 11), (20, 255, 3, 12)])
def test_terms_against_polynomial_expansion(M, nmax, nt, seed):
    rng = np.random.default_rng(seed)
    prior, S, w_all, pdw = random_case(rng, M, nmax, nt)
    got = O.cphd_terms(prior, S, w_all, pdw, BW, LAM, KAP)
    ref = brute(prior, S, w_all, pdw, BW, LAM, KAP)
    # fp32 log-domain recursion over M steps with terms of magnitude ~1e3: 2e-3 absolute in the logs
    assert np.allclose(got["lz"], ref["lz"], atol=2e-3)
    assert abs(got["r1"] - ref["r1"]) < 2e-3 * max(1.0, ref["r1"])
    assert abs(got["lY0"] - ref["lY0"]) < 1e-4 * abs(ref["lY0"]) + 2e-3
    live = ref["cn"] > -40
    assert np.allclose(got["cn"][live], ref["cn"][live], atol=5e-3)
    assert abs(np.log(np.exp(got["cn"].astype(np.float64)).sum())) < 2e-3


@pytest.mark.parametrize("M,nmax,nt,seed", [(9, 30, 4, 20), (64, 255, 40, 21), (130, 255, 90, 22)])
def test_prefix_suffix_form_equals_leave_one_out_recursions(M, nmax, nt, seed):
    """<Y1[Z\\m],p> from the O(M^2) prefix/suffix recursion (what the device runs) against M separate ESF
    recursions (the .bak's O(M^3) structure): same numbers up to fp32 rounding of different sum orders"""
    rng = np.random.default_rng(seed)
    prior, S, w_all, pdw = random_case(rng, M, nmax, nt)
    fast = O.cphd_terms(prior, S, w_all, pdw, BW, LAM, KAP)
    O.cphd_set_reference_esf(True)
    try:
        ref = O.cphd_terms(prior, S, w_all, pdw, BW, LAM, KAP)
    finally:
        O.cphd_set_reference_esf(False)
    assert np.allclose(fast["lz"], ref["lz"], atol=3e-4)
    assert fast["r1"] == ref["r1"] and fast["lY0"] == ref["lY0"] and np.array_equal(fast["cn"], ref["cn"])


def test_poisson_prior_reduces_to_the_phd_update():
    """CPHD with a Poisson cardinality of mean <1,v> is the PHD filter: weights, missed-detection
    factor 1 and the particle weight up to the particle-independent constant M log(lambda/kappa) - lambda"""
    cfg = O.default_config()
    rng = np.random.default_rng(3)
    G, M = 24, 12
    pose = np.zeros(1, O.POSE)
    gmap = np.zeros(G, O.GAUSSIAN)
    ang = rng.uniform(-np.pi, np.pi, G)
    rad = rng.uniform(2, 13, G)
    gmap["mean"] = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
    gmap["cov"][:, 0] = gmap["cov"][:, 3] = 0.04
    gmap["weight"] = rng.uniform(0.3, 1.0, G)
    gmap["mean"][:3] *= 3.0                                                   # three features out of range (pD = 0)
    z = np.zeros(M, O.MEAS)
    for m in range(M):
        g = gmap[4 + m % 10]
        z["range"][m] = np.hypot(*g["mean"]) + rng.normal(0, 0.1)
        z["bearing"][m] = np.arctan2(g["mean"][1], g["mean"][0]) + rng.normal(0, 0.005)
    lam = 20.0
    w1 = float(gmap["weight"].sum()) + M * cfg.birthWeight
    # the births' binomial cardinality is convolved in by the update: start from Poisson(<1,map>) and the
    # predicted law is Poisson(<1,v>) up to O(M bw^2)
    prior = log_poisson(float(gmap["weight"].sum()), 255)
    c = O.cphd_update_particle(pose[0], gmap, z, cfg, lam, prior)
    p = O.update_particle(pose[0], gmap, z, cfg)
    assert abs(c["r1"] - 1.0) < 1e-3
    assert len(c["map"]) == len(p["map"])
    assert np.allclose(c["map"]["weight"], p["map"]["weight"], rtol=3e-3, atol=1e-6)
    assert np.allclose(c["map"]["mean"], p["map"]["mean"], atol=1e-3)
    const = M * (np.log(lam) - np.log(cfg.clutterDensity)) - lam
    assert abs(c["dlogw"] - (p["dlogw"] + const)) < 2e-2
    # posterior cardinality: normalised, mean close to the posterior mass of the map
    pn = np.exp(c["cn"].astype(np.float64))
    assert abs(pn.sum() - 1) < 1e-3
    assert abs((pn * np.arange(256)).sum() - float(c["map"]["weight"].sum())) < 0.05 * w1


def test_informative_prior_changes_the_update():
    """a sharp cardinality prior (exactly n targets) is NOT the PHD filter: the missed-detection factor
    leaves 1 and the posterior cardinality stays concentrated"""
    rng = np.random.default_rng(8)
    M, nmax, nt = 10, 63, 6
    _, S, w_all, pdw = random_case(rng, M, nmax, nt)
    prior = np.full(nmax + 1, -80.0, np.float32)
    prior[nt] = 0.0
    got = O.cphd_terms(prior, S, w_all, pdw, BW, LAM, KAP)
    ref = brute(prior, S, w_all, pdw, BW, LAM, KAP)
    assert abs(got["r1"] - ref["r1"]) < 2e-3 * max(1.0, ref["r1"]) and abs(ref["r1"] - 1.0) > 0.02
    pn = np.exp(got["cn"].astype(np.float64))
    assert pn[nt - 1:nt + 2].sum() > 0.99
