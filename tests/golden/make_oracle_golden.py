#!/usr/bin/env python3
"""Freeze per-stage outputs of the CPU oracle as golden binaries (SURVEY.md §8c "Fixtures to commit"; the idea of
the reference's own state100.bin snapshot, src/main.cpp:1262-1269,1314-1321).

    python3 tests/golden/make_oracle_golden.py        # writes tests/golden/oracle_steps.npz

3 seeds x {1 x 64 x 32, 8 x 64 x 32, 4 x 256 x 64 (clustered landmarks)}: inputs (poses, log-weights, maps, measurement
set, control noise, resampling uniform), predicted poses, per-particle survivors (pruned update components + slab
indices), merged maps, log-weight increments, decision margins, normalised weights, resampling indices, and (round 5) the
unpruned slab as far as an admissible device survivor can index it (`slab_keep`, `slab_keep_idx`), the class counts and the
float64 log-weight increment — what tests/parity_utils.compare_particle_with_oracle needs to hold the device to this file
with its first-order proofs.
`out_maps` are the merged maps under the exact, order-free moment sums (o_config.mergeSums = 0: what the device computes);
`out_maps_float` the same merge with float sums in weight order (mergeSums = 1) — bit for bit the `out_maps` of the file as
it stood before round 3 (checked when the file was regenerated).

Both the oracle (tests/test_oracle_golden.py, CPU) and the device (tests/test_gpu_golden.py) are tested against
this FILE, not against each other only — so the oracle and the kernels cannot drift together unnoticed.  The file
is regenerated only by a deliberate, reviewed change of the oracle's semantics (say so in the commit).
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [(n, g, m, seed) for (n, g, m) in ((1, 64, 32), (8, 64, 32), (4, 256, 64)) for seed in (101, 202, 303)]
CONTROL = (2.0, 0.05)   # (v_encoder, alpha)


def case_key(n, g, m, seed):
    return "n%d_g%d_m%d_s%d" % (n, g, m, seed)


def make_case(n, g, m, seed):
    from oracle import oracle as O
    from parity_utils import oracle_config_from, oracle_full_update
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    w = S.make_workload(n, g, m, seed=seed, clustered=g >= 256)
    cfg = P.default_config()
    ocfg = oracle_config_from(cfg)
    z = w["z"][0]
    pred = O.predict_ackerman(w["poses"], CONTROL[1], CONTROL[0], w["noise"][0], ocfg)
    ocfg_float = oracle_config_from(cfg, mergeSums=1)       # the merge with float sums in weight order (round 2's definition)
    surv, sidx, nsurv, maps, maps_float, sizes, dlogw, margins = [], [], [], [], [], [], [], []
    keep, keep_idx, nkeep, counts, extra = [], [], [], [], []
    for p in range(n):
        r = oracle_full_update(pred[p], w["maps"][p, :w["sizes"][p]], z, ocfg)
        surv.append(r["survivors"]); sidx.append(r["slab_idx"]); nsurv.append(len(r["survivors"]))
        maps.append(r["map"]); sizes.append(len(r["map"])); dlogw.append(r["dlogw"])
        margins.append((r["prune_margin"], r["margin"][0], r["margin"][1]))
        # round 5: the UNPRUNED slab, as far as any admissible device survivor can index it — every update component whose
        # weight reaches half the prune threshold (a member only one side keeps must be within 5e-4 of it) and all nearly-in-range
        # features — so that the device is held to THIS FILE by the proof machinery of parity_utils.compare_particle_with_oracle
        # (the oracle's own values under the device's decisions), not by fixed margins
        sa, nu = r["slab_all"], r["n_update"]
        k = np.flatnonzero((np.arange(len(sa)) >= nu) | (sa["weight"] >= 0.5 * ocfg.minFeatureWeight)).astype(np.int32)
        assert np.isin(r["slab_idx"], k).all()
        keep.append(sa[k]); keep_idx.append(k); nkeep.append(len(k))
        counts.append((r["n_in"], len(sa) - nu, int((r["cls"] == 0).sum()), len(sa)))
        extra.append((r["card"], r["dlogw_f64"]))
        rf = oracle_full_update(pred[p], w["maps"][p, :w["sizes"][p]], z, ocfg_float)
        assert len(rf["map"]) == len(r["map"])             # the two definitions differ in roundings, never in structure
        maps_float.append(rf["map"])
    dlogw = np.array(dlogw, np.float32)
    lw = O.normalize_weights(w["logw"], dlogw)
    idx = O.resample(lw, w["uniform"][0])
    return {
        "poses": w["poses"], "logw": w["logw"], "maps": w["maps"], "sizes": w["sizes"], "z": z, "noise": w["noise"][0],
        "uniform": np.float64(w["uniform"][0]), "pred": pred,
        "surv": np.concatenate(surv), "sidx": np.concatenate(sidx), "nsurv": np.array(nsurv, np.int32),
        "out_maps": np.concatenate(maps), "out_maps_float": np.concatenate(maps_float),
        "out_sizes": np.array(sizes, np.int32), "dlogw": dlogw,
        "margins": np.array(margins, np.float64), "logw_norm": lw, "neff": np.float32(O.neff(lw)), "idx": idx,
        # (added in round 5; everything above is bit for bit the round-3 file)
        "slab_keep": np.concatenate(keep), "slab_keep_idx": np.concatenate(keep_idx), "nkeep": np.array(nkeep, np.int32),
        "counts": np.array(counts, np.int32),                  # per particle: n_in, n_near, n_out0, len(slab_all)
        "card_f64": np.array(extra, np.float64),               # per particle: predicted cardinality, float64 log-weight increment
    }


def main():
    out = {}
    for (n, g, m, seed) in CASES:
        c = make_case(n, g, m, seed)
        for k, v in c.items():
            out[case_key(n, g, m, seed) + "/" + k] = v
        print(case_key(n, g, m, seed), "survivors", c["nsurv"].tolist(), "map sizes", c["out_sizes"].tolist())
    path = os.path.join(HERE, "oracle_steps.npz")
    if os.path.exists(path):
        # a regeneration that only ADDS arrays must leave every existing array bit-identical
        old = np.load(path)
        changed = [k for k in old.files if k not in out or not (old[k].dtype == out[k].dtype and old[k].shape == out[k].shape
                                                                  and old[k].tobytes() == out[k].tobytes())]
        print("arrays of the existing file that would change:", changed or "none")
        if changed and "--allow-changes" not in sys.argv:
            raise SystemExit("refusing to overwrite: pass --allow-changes for a deliberate change of the oracle's semantics")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
