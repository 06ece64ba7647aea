"""The CPU oracle against its own FROZEN outputs (tests/golden/oracle_steps.npz; SURVEY.md §8c fixtures): the oracle was
changed during round 1 (fixed-point CDF, det_exp) while every parity test compared the device with the LIVE oracle —
nothing would have caught the two drifting together.  The device is tested against the same file
(tests/test_gpu_golden.py).

Stages without transcendentals (merge, resampling indices from given weights) must reproduce the file BIT FOR BIT on
any host; the stages that call libm (atan2f, logf, expf: last-bit rounding is a property of the host's libm) within
the tolerances of the device tests, structure compared where the frozen decision margins allow."""
import numpy as np
import pytest

from golden_utils import CASES, CONTROL, load_case
from oracle import oracle as O
from parity_utils import (OBS, assert_maps_close, compare_particle_with_oracle, dlogw_tolerance, oracle_config_from,
                          oracle_full_update, pkg)



@pytest.mark.parametrize("n,g,m,seed", CASES)
def test_oracle_reproduces_its_frozen_outputs(n, g, m, seed):
    c = load_case(n, g, m, seed)
    ocfg = oracle_config_from(pkg().default_config())
    ocfg_float = oracle_config_from(pkg().default_config(), mergeSums=1)
    pred = O.predict_ackerman(c["poses"], CONTROL[1], CONTROL[0], c["noise"], ocfg)
    for k in ("px", "py", "ptheta"):
        assert np.abs(pred[k] - c["pred"][k]).max() < 2e-6, k
    n_struct = 0
    dl = []
    OBS.clear()
    for p in range(n):
        # the update from the FROZEN predicted pose (so a libm difference in the predict does not leak in)
        r = oracle_full_update(c["pred"][p], c["maps"][p, :c["sizes"][p]], c["z"], ocfg)
        dl.append(r["dlogw"])
        assert abs(r["dlogw"] - c["dlogw"][p]) < dlogw_tolerance(c["dlogw"][p], m, r["card"], r["n_in"])
        # merge: no transcendental in it -> bit for bit on the frozen survivors
        om = O.merge(c["surv_of"](p), ocfg)
        cls0 = c["maps"][p, :c["sizes"][p]][r["cls"] == 0]
        want = c["map_of"](p)
        got = np.concatenate([om, cls0]) if len(cls0) else om
        assert len(got) == len(want)
        for fld in ("weight", "mean", "cov"):
            assert np.array_equal(got[fld].view(np.uint32), want[fld].view(np.uint32)), (p, fld)
        # the same merge with float sums in weight order (o_config.mergeSums = 1, round 2's definition): frozen too, and
        # within rounding of the exact sums
        omf = O.merge(c["surv_of"](p), ocfg_float)
        wantf = c["map_float_of"](p)
        gotf = np.concatenate([omf, cls0]) if len(cls0) else omf
        assert len(gotf) == len(wantf) == len(want)
        for fld in ("weight", "mean", "cov"):
            assert np.array_equal(gotf[fld].view(np.uint32), wantf[fld].view(np.uint32)), (p, fld)
        assert np.allclose(gotf["weight"], got["weight"], rtol=2e-6, atol=0)
        assert np.abs(gotf["mean"] - got["mean"]).max() <= 2e-6 * max(1.0, np.abs(got["mean"]).max())
        assert np.allclose(gotf["cov"], got["cov"], rtol=3e-5, atol=1e-9)
        # the live oracle (this host's libm) against the FILE through the same proof machinery the device goes through
        # (tests/test_gpu_golden.py): every particle, no margins; on the host that wrote the file the two are identical
        ref = c["ref_of"](p)
        assert ref["n_in"] == r["n_in"] and len(ref["slab_all"]) == len(r["slab_all"])
        kept = ref["slab_all"]["weight"] != 0
        assert np.array_equal(ref["slab_all"][kept], r["slab_all"][kept]) or np.allclose(ref["slab_all"]["weight"][kept],
                                                                                         r["slab_all"]["weight"][kept], rtol=5e-4)
        res = compare_particle_with_oracle(r["map"], r["survivors"], r["slab_idx"], ref, ocfg, m, dlw=r["dlogw"],
                                           what="live oracle vs file, particle %d" % p)
        n_struct += bool(res["structural"])
    assert OBS.count.get("hellinger_marginal_particles", 0) == 0 and OBS.count.get("nan_distance_particles", 0) == 0
    # weights from the frozen increments; indices from the frozen weights: exact integer arithmetic (fixed-point CDF)
    lw = O.normalize_weights(c["logw"], c["dlogw"])
    assert np.abs(lw - c["logw_norm"]).max() < 2e-6
    assert abs(O.neff(c["logw_norm"]) - float(c["neff"])) < 1e-5 * max(1.0, float(c["neff"]))
    assert np.array_equal(O.resample(c["logw_norm"], float(c["uniform"])), c["idx"])


def test_frozen_file_covers_the_three_shapes_and_seeds():
    for (n, g, m, seed) in CASES:
        c = load_case(n, g, m, seed)
        assert c["poses"].shape == (n,) and c["maps"].shape == (n, g) and c["z"].shape == (m,)
        assert c["nsurv"].sum() == len(c["surv"]) and c["out_sizes"].sum() == len(c["out_maps"])
        assert c["nkeep"].sum() == len(c["slab_keep"]) == len(c["slab_keep_idx"]) and c["counts"].shape == (n, 4)
        for p in range(n):
            ref = c["ref_of"](p)
            assert np.array_equal(ref["slab_all"][ref["slab_idx"]], ref["survivors"])       # the survivors ARE slab entries
        assert np.all(np.diff(c["idx"]) >= 0)
