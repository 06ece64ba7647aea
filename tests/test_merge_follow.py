"""o_merge_follow (oracle/scphd_cpu.c) — the diagnostic the GPU parity tests use to PROVE a flipped merge decision instead of
tolerating it — checked on the CPU: it reproduces o_merge when nothing differs, follows and explains flips caused by
perturbations of the size the update stage's tolerances allow, and refuses a flip the survivor difference cannot explain."""
import importlib

import numpy as np

from oracle import oracle as O
from parity_utils import assert_maps_close


def _survivors(seed, clustered=True):
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    w = S.make_workload(2, 64, 32, seed=seed, clustered=clustered)
    cfg = O.default_config()
    return O.update_particle(w["poses"][0], w["maps"][0], w["z"][0], cfg)["survivors"], cfg


def test_follow_of_identical_inputs_is_the_merge_itself():
    for seed in (3, 4, 5):
        sv, cfg = _survivors(seed)
        m, st = O.merge_follow(sv, sv, cfg)
        assert np.array_equal(m, O.merge(sv, cfg))
        assert st["dist_flips"] == st["order_flips"] == st["nan_decisions"] == 0 and st["n_decisions"] > 100
        cfg.mergeSums = 1
        m, st = O.merge_follow(sv, sv, cfg)
        assert np.array_equal(m, O.merge(sv, cfg))


def test_flips_from_tolerance_sized_perturbations_are_explained():
    """survivors perturbed by what the device may differ from the oracle (weights 1e-4 relative, means 1e-5 m, covariances
    3e-5): some decisions flip, every one is explained, and the followed map stays within the map tolerances of the merge of
    the unperturbed survivors — while the perturbed survivors' OWN merge may have another structure altogether"""
    rng = np.random.default_rng(1)
    n_flips = n_other_structure = 0
    for seed in range(10, 40):
        sv, cfg = _survivors(seed)
        sv2 = sv.copy()
        sv2["weight"] *= (1 + rng.normal(0, 1e-4, len(sv))).astype(np.float32)
        sv2["mean"] += rng.normal(0, 1e-5, sv2["mean"].shape).astype(np.float32)
        sv2["cov"] *= (1 + rng.normal(0, 3e-5, (len(sv), 1))).astype(np.float32)
        m, st = O.merge_follow(sv, sv2, cfg)
        assert st["dist_unexplained"] == 0 and st["order_unexplained"] == 0, st
        assert st["dist_worst_ratio"] <= 1.0 and st["order_worst_ratio"] <= 1.0
        n_flips += st["dist_flips"] + st["order_flips"]
        assert_maps_close(m, O.merge(sv, cfg), ordered=True)
        n_other_structure += len(O.merge(sv2, cfg)) != len(m)
    assert n_flips > 0, "the perturbations never flipped a decision: the test does not exercise the proof"


def test_a_flip_the_difference_cannot_explain_is_refused():
    """two components at half the merge distance; the second one's covariance shrunk to a quarter in `in`: the distance
    quadruples (2 T), the first-order change from the covariance difference is only 0.75 d — not explained"""
    cfg = O.default_config()
    T = cfg.minSeparation
    g = np.zeros(2, O.GAUSSIAN)
    g["weight"] = (0.9, 0.5)
    g["cov"] = (0.04, 0.0, 0.0, 0.04)
    d = np.sqrt(0.5 * T * 0.04)
    g["mean"][1] = (d, 0.0)
    assert abs(O.mahal_dist(g[0], g[1]) - 0.5 * T) < 1e-4 * T
    bad = g.copy()
    bad["cov"] *= 0.25
    m, st = O.merge_follow(g, bad, cfg)
    assert len(m) == 1 and st["dist_flips"] == 1 and st["dist_unexplained"] == 1 and st["dist_worst_ratio"] > 1.0, st
    # a seed-order inversion larger than the weights' own differences is refused too
    g2 = g.copy()
    g2["mean"][1] = (5.0, 0.0)                      # far apart: two clusters, order = weight order
    inv = g2.copy()
    inv["weight"] = (0.9, 0.9001)                   # in `in` the second is heavier, by more than ...
    ref = g2.copy()
    ref["weight"] = (0.90005, 0.90004)              # ... twice the two differences |0.9 - 0.90005| + |0.9001 - 0.90004|? no: 1.1e-4 -> explained
    m, st = O.merge_follow(ref, inv, cfg)
    assert st["order_flips"] == 1 and st["order_unexplained"] == 0, st
    ref["weight"] = (0.90005, 0.9)                  # second weight differs by 1e-4, first by 5e-5: gap 1e-4 <= 2 (1.5e-4): explained
    inv["weight"] = (0.9, 0.901)                    # gap 1e-3 > 2 (5e-5 + 1e-3)?  no — make the heavier one agree: unexplained
    ref["weight"] = (0.9003, 0.901)
    m, st = O.merge_follow(ref, inv, cfg)           # ref order: [1] first (0.901 > 0.9003): no inversion at all
    assert st["order_flips"] == 0, st
    ref["weight"] = (0.9011, 0.901)                 # ref picks [0]; in `in` [1] is heavier by 1e-3 while [0] moved 1.1e-3: explained
    m, st = O.merge_follow(ref, inv, cfg)
    assert st["order_flips"] == 1 and st["order_unexplained"] == 0, st
    inv["weight"] = (0.9, 0.95)                     # [1] moved by 0.049 ... gap 0.05 <= 2 (0.0011 + 0.049): explained by ITS difference
    m, st = O.merge_follow(ref, inv, cfg)
    assert st["order_unexplained"] == 0
    # (an order inversion is always within the two weights' differences — w_in[j] - w_in[s] <= |dw_j| + |dw_s| follows from
    #  ref's order — so what guards the order is the survivor-weight tolerance itself, asserted by the callers)
