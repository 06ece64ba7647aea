"""The device against the FROZEN oracle outputs of tests/golden/oracle_steps.npz (3 seeds x {1x64x32, 8x64x32, 4x256x64}):
predict, survivors, merged maps, log-weight increments, normalised weights, resampling indices — through the C-ABI.
The companion tests/test_oracle_golden.py holds the live oracle to the same file."""
import numpy as np
import pytest

from golden_utils import CASES, CONTROL, load_case
from parity_utils import assert_maps_close, dlogw_tolerance, pkg

pytestmark = pytest.mark.gpu

PRUNE_MARGIN = 2e-3
MERGE_MARGIN = 2e-4


@pytest.mark.parametrize("n,g,m,seed", CASES)
def test_device_reproduces_the_frozen_oracle_outputs(n, g, m, seed):
    P = pkg()
    c = load_case(n, g, m, seed)
    cfg = P.default_config()
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
            card = 0.95 * float(c["maps"][p, :c["sizes"][p]]["weight"].sum()) + m * 1e-4    # predicted cardinality (upper bound)
            assert abs(dlw[p] - c["dlogw"][p]) < dlogw_tolerance(c["dlogw"][p], m, card, g), (p, dlw[p], c["dlogw"][p])
            pm, mm = c["margins"][p, 0], c["margins"][p, 1]
            if pm > PRUNE_MARGIN:
                surv, sidx = f.survivors(p)
                assert np.array_equal(sidx, c["sidx_of"](p)), "particle %d: survivor set differs from the frozen one" % p
                assert_maps_close(surv, c["surv_of"](p), ordered=True, what="survivors of particle %d" % p)
                if mm > MERGE_MARGIN:
                    n_struct += 1
                    assert_maps_close(maps[p], c["map_of"](p), what="map of particle %d" % p)
        assert np.abs(lw - c["logw_norm"]).max() < 2e-3
        # resampling from the FROZEN normalised weights: bit-exact indices (fixed-point CDF)
        f.set_particles(None, c["logw_norm"])
        assert abs(f.neff() - float(c["neff"])) < 1e-5 * max(1.0, float(c["neff"]))
        idx = f.resample(float(c["uniform"]))
        assert np.array_equal(idx, c["idx"])
    assert n_struct >= 0.5 * n, (n_struct, n)
