"""The device against the FROZEN oracle outputs of tests/golden/oracle_steps.npz (3 seeds x {1x64x32, 8x64x32, 4x256x64}):
predict, survivors, merged maps, log-weight increments, normalised weights, resampling indices — through the C-ABI.
The companion tests/test_oracle_golden.py holds the live oracle to the same file."""
import numpy as np
import pytest

from golden_utils import CASES, CONTROL, load_case
from parity_utils import OBS, compare_particle_with_oracle, oracle_config_from, pkg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,g,m,seed", CASES)
def test_device_reproduces_the_frozen_oracle_outputs(n, g, m, seed):
    """EVERY particle of the file (39 in all) goes through parity_utils.compare_particle_with_oracle with the FILE as the
    oracle's side: log-weight increment, merge stage bit for bit, survivor set (a member only one side keeps must be marginal),
    and the device's map cluster by cluster against the file's values merged under the device's decisions — every decision the
    file's values alone would take differently proven by its first-order sensitivity.  (Rounds 1-4 compared structure only
    where fixed margins of the file said "clear", and let half of the particles go uncompared.)"""
    P = pkg()
    c = load_case(n, g, m, seed)
    cfg = P.default_config()
    ocfg = oracle_config_from(cfg)
    OBS.clear()
    n_struct = 0
    with P.PhdFilter(cfg, n_particles=n, map_capacity=2 * g, max_measurements=m) as f:
        f.set_particles(c["poses"], c["logw"])
        f.set_maps(c["maps"], c["sizes"])
        f.debug(True)
        f.predict(CONTROL, c["noise"])
        poses, _ = f.get_particles()
        for k in ("px", "py", "ptheta"):
            assert np.abs(poses[k] - c["pred"][k]).max() < 2e-6, k
        f.update(c["z"])
        f.status()
        maps = f.get_maps()
        dlw = f.weight_increments()
        _, lw = f.get_particles()
        for p in range(n):
            surv, sidx = f.survivors(p)
            r = compare_particle_with_oracle(maps[p], surv, sidx, c["ref_of"](p), ocfg, m, dlw=dlw[p],
                                             what="golden %dx%dx%d seed %d particle %d" % (n, g, m, seed, p))
            n_struct += bool(r["structural"])
        assert np.abs(lw - c["logw_norm"]).max() < 2e-3
        # resampling from the FROZEN normalised weights: bit-exact indices (fixed-point CDF)
        f.set_particles(None, c["logw_norm"])
        assert abs(f.neff() - float(c["neff"])) < 1e-5 * max(1.0, float(c["neff"]))
        idx = f.resample(float(c["uniform"]))
        assert np.array_equal(idx, c["idx"])
    # Mahalanobis metric: no particle may leave the comparison early (Hellinger-marginal / NaN distances do not exist here)
    assert OBS.count.get("hellinger_marginal_particles", 0) == 0 and OBS.count.get("nan_distance_particles", 0) == 0, OBS.report()
    print("\n  golden %dx%dx%d seed %d: %d / %d particles with the file's own clusters; %s" % (n, g, m, seed, n_struct, n, OBS.report()))
