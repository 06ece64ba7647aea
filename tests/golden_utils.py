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
    ko = np.concatenate([[0], np.cumsum(c["nkeep"])])

    def ref_of(p):
        """the oracle's side of parity_utils.compare_particle_with_oracle for particle p, FROM THE FILE: its survivors, slab
        indices, merged map, log-weight increment (+ the float64 one, the predicted cardinality) and the unpruned slab —
        stored as far as an admissible device survivor can index it (weights >= half the prune threshold + the nearly-in-range
        features); everything else reads weight 0, which no survivor may have"""
        n_in, n_near, n_out0, n_slab = (int(v) for v in c["counts"][p])
        slab_all = np.zeros(n_slab, c["slab_keep"].dtype)
        slab_all[c["slab_keep_idx"][ko[p]:ko[p + 1]]] = c["slab_keep"][ko[p]:ko[p + 1]]
        fmap = c["map_of"](p)
        return {"dlogw": float(c["dlogw"][p]), "card": float(c["card_f64"][p, 0]), "dlogw_f64": float(c["card_f64"][p, 1]),
                "n_in": n_in, "out0": fmap[len(fmap) - n_out0:], "slab_idx": c["sidx_of"](p), "slab_all": slab_all,
                "survivors": c["surv_of"](p), "map": fmap}
    c["ref_of"] = ref_of
    return c
