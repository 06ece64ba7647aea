"""Access to the frozen oracle outputs of tests/golden/oracle_steps.npz (written by tests/golden/make_oracle_golden.py)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(n, g, m, seed) for (n, g, m) in ((1, 64, 32), (8, 64, 32), (4, 256, 64)) for seed in (101, 202, 303)]
CONTROL = (2.0, 0.05)
_cache = {}


def load_case(n, g, m, seed):
    if "z" not in _cache:
        _cache["z"] = np.load(os.path.join(HERE, "golden", "oracle_steps.npz"))
    z = _cache["z"]
    pre = "n%d_g%d_m%d_s%d/" % (n, g, m, seed)
    c = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    assert c, "golden case missing: " + pre
    so = np.concatenate([[0], np.cumsum(c["nsurv"])])
    mo = np.concatenate([[0], np.cumsum(c["out_sizes"])])
    c["surv_of"] = lambda p: c["surv"][so[p]:so[p + 1]]
    c["sidx_of"] = lambda p: c["sidx"][so[p]:so[p + 1]]
    c["map_of"] = lambda p: c["out_maps"][mo[p]:mo[p + 1]]
    c["map_float_of"] = lambda p: c["out_maps_float"][mo[p]:mo[p + 1]]
    return c
