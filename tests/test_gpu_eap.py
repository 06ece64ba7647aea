"""SURVEY.md §8(f) N2 on hardware: the expected-a-posteriori map (computeExpectedMap,
src/main.cpp:290-316) and the device-wide reduceGaussianMixture (src/gm_reduce.cpp:57-134) against
the oracle's literal transcription (o_gm_reduce / o_expected_map).

The reduction contains no transcendental: sqrt, divide, multiply and add are correctly rounded on
both sides and the moment sums are taken in the reference's order, so the bar is BIT-EXACT for
every output field, in the reference's output order."""
import time

import numpy as np
import pytest

from oracle import oracle as O
from parity_utils import pkg, synthetic

pytestmark = pytest.mark.gpu


def assert_bit_equal(got, ref, what=""):
    assert len(got) == len(ref), "%s: %d vs %d Gaussians" % (what, len(got), len(ref))
    for fld in ("weight", "mean", "cov"):
        a, b = got[fld].view(np.uint32), ref[fld].view(np.uint32)
        if not np.array_equal(a, b):
            # NaN payloads may differ; values must not
            assert np.array_equal(got[fld], ref[fld], equal_nan=True), \
                "%s: %s differs (max %g)" % (what, fld, np.nanmax(np.abs(got[fld] - ref[fld])))


def mixture(rng, n, n_centres, spread=0.15, extent=40.0, asym=False, ties=False):
    P = pkg()
    c = rng.uniform(-extent, extent, (n_centres, 2))
    g = np.zeros(n, P.GAUSSIAN)
    k = rng.integers(0, n_centres, n)
    g["mean"] = (c[k] + spread * rng.standard_normal((n, 2))).astype(np.float32)
    s = rng.uniform(0.05, 0.4, (n, 2))
    rho = rng.uniform(-0.9, 0.9, n)
    g["cov"][:, 0] = s[:, 0] ** 2
    g["cov"][:, 3] = s[:, 1] ** 2
    g["cov"][:, 1] = g["cov"][:, 2] = rho * s[:, 0] * s[:, 1]
    if asym:
        g["cov"][:, 2] *= rng.uniform(0.9, 1.1, n).astype(np.float32)
    g["weight"] = rng.uniform(1e-4, 1.0, n).astype(np.float32)
    if ties:
        g["weight"] = np.round(g["weight"] * 8) / 8 + np.float32(0.125)
    return g


@pytest.fixture(scope="module")
def filt():
    P = pkg()
    with P.PhdFilter(P.default_config(), n_particles=4, map_capacity=64, max_measurements=8) as f:
        yield f


@pytest.mark.parametrize("n,centres,asym,ties", [(1, 1, False, False), (2, 1, False, False), (63, 5, False, False),
                                                 (64, 64, False, False), (65, 3, True, False), (200, 200, False, False),
                                                 (3000, 40, False, False), (3000, 40, True, True), (20000, 150, False, False)])
def test_gm_reduce_bit_exact(filt, n, centres, asym, ties):
    rng = np.random.default_rng(1000 + n + centres)
    g = mixture(rng, n, centres, asym=asym, ties=ties)
    ref = O.gm_reduce(g, 10.0)
    got = filt.gm_reduce(g, 10.0)
    assert_bit_equal(got, ref, "n=%d" % n)
    assert filt.gm_rounds() >= 1
    assert abs(float(got["weight"].astype(np.float64).sum()) - float(g["weight"].astype(np.float64).sum())) < 1e-3 * n


def test_gm_reduce_edge_cases(filt):
    P = pkg()
    assert len(filt.gm_reduce(np.zeros(0, P.GAUSSIAN), 10.0)) == 0
    rng = np.random.default_rng(5)
    # nothing merges (far apart, tiny threshold): K == n, many rounds
    g = mixture(rng, 300, 300)
    g["mean"] = rng.uniform(-1e4, 1e4, (300, 2)).astype(np.float32)
    got = filt.gm_reduce(g, 1e-6)
    assert_bit_equal(got, O.gm_reduce(g, 1e-6), "no merges")
    assert len(got) == 300 and filt.gm_rounds() == 5
    # everything merges into one
    g = mixture(rng, 1000, 1, spread=0.01)
    got = filt.gm_reduce(g, 1e6)
    assert_bit_equal(got, O.gm_reduce(g, 1e6), "one cluster")
    assert len(got) == 1 and filt.gm_rounds() == 1
    # exact duplicates (resampled particles carry identical maps) and degenerate covariances
    g = mixture(rng, 500, 20)
    g = np.concatenate([g, g, g[:100]])
    g["cov"][7] = 0
    g["cov"][11, 0] = -1.0
    g["mean"][13] = np.inf
    ref = O.gm_reduce(g, 10.0)
    got = filt.gm_reduce(g, 10.0)
    assert_bit_equal(got, ref, "duplicates/degenerate")


def test_expected_map_matches_oracle():
    P, S = pkg(), synthetic()
    cfg = P.default_config()
    w = S.make_workload(48, 24, 10, seed=21, n_meas_sets=4)
    with P.PhdFilter(cfg, n_particles=48, map_capacity=96, max_measurements=16) as f:
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        for k in range(4):
            f.predict((2.0, 0.02), None)
            f.update(w["z"][k])
            if k == 2:
                f.resample(0.37)          # duplicates: several particles now share one map
                maps = f.get_maps()
                _, logw = f.get_particles()
                sizes = np.array([len(m) for m in maps], np.int32)
                ref = O.expected_map(np.concatenate(maps), sizes, logw, cfg.minSeparation)
                assert_bit_equal(f.expected_map(), ref, "after resample")
        maps = f.get_maps()
        _, logw = f.get_particles()
        sizes = np.array([len(m) for m in maps], np.int32)
        ref = O.expected_map(np.concatenate(maps), sizes, logw, cfg.minSeparation)
        got = f.expected_map()
        assert_bit_equal(got, ref, "expected map")
        # the EAP map keeps the PHD mass: sum_k w_k = sum_p exp(logw_p) * sum_i w_pi
        mass = sum(float(np.exp(np.float64(lw))) * float(m["weight"].astype(np.float64).sum()) for lw, m in zip(logw, maps))
        assert abs(float(got["weight"].astype(np.float64).sum()) - mass) < 1e-3 * max(mass, 1.0)
        # capacity error reports the size needed
        import ctypes as C
        n = C.c_int32(0)
        out = np.zeros(1, P.GAUSSIAN)
        rc = P._lib.lib().phd_expected_map(f._h, P._lib.ptr(out), 1, C.byref(n))
        assert rc == -5 and n.value == len(ref)


def test_expected_map_empty_maps():
    P = pkg()
    with P.PhdFilter(P.default_config(), n_particles=8, map_capacity=16, max_measurements=8) as f:
        assert len(f.expected_map()) == 0                                        # "no features" (src/main.cpp:308-313)


@pytest.mark.parametrize("cfg_id", [2, 3])
def test_expected_map_full_size(cfg_id):
    """BASELINE configs[1] (256 particles x 64 features) and configs[2] (4096 x 256, ~10^6 Gaussians):
    parity + a timing line against the oracle"""
    P, S = pkg(), synthetic()
    cfg = P.default_config()
    w = S.config_workload(cfg_id)
    with P.PhdFilter(cfg, n_particles=w["N"], map_capacity=2 * w["G"], max_measurements=64) as f:
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        f.update(w["z"][0])
        f.expected_map()                                                          # allocate / warm up
        t0 = time.perf_counter()
        got = f.expected_map()
        t_gpu = time.perf_counter() - t0
        maps = f.get_maps()
        _, logw = f.get_particles()
        sizes = np.array([len(m) for m in maps], np.int32)
        cat = np.concatenate(maps)
        t0 = time.perf_counter()
        ref = O.expected_map(cat, sizes, logw, cfg.minSeparation)
        t_cpu = time.perf_counter() - t0
        assert_bit_equal(got, ref, "config %d" % cfg_id)
        print("\nEAP map, %d Gaussians -> %d: device %.2f ms (%d rounds), oracle %.1f ms"
              % (len(cat), len(got), 1e3 * t_gpu, f.gm_rounds(), 1e3 * t_cpu))
