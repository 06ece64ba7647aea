#!/usr/bin/env python3
"""Randomised sweep of the update kernel's FAST PATH (the instantiations with a BASELINE.json layout, a full scan and — PHD — the
Mahalanobis metric compiled in; csrc/phd_kernels.hip, LAYOUT) against the general instantiations, bit for bit: the same random
workload through a default filter and through one created with PHD_LAYOUT=0 — staged update (maps, log-weight increments) and
the fused single-launch step (maps, poses, weights after the resample), PHD and CPHD, both layouts, particle counts on both sides
of the three-per-CU threshold and of the block-form tail, random map sizes, clustered and scattered maps.

    python tools/fuzz_layout.py [seconds=120] [first_seed=1]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    L = P._lib.lib()
    dev = torch.device("cuda:0")
    t0 = time.time()
    n_ok = n_fail = n_fast = n_skip = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        layout = int(rng.choice([1, 1, 2]))
        cap, M = (512, 64) if layout == 1 else (128, 32)
        ft = int(rng.random() < 0.35) if layout == 1 else 0
        N = int(rng.choice([600, 900, 900, 4200, 64] if layout == 1 else [1, 17, 256, 500]))
        G = int(rng.integers(1, cap // 2 + 1))
        clustered = bool(rng.integers(0, 2))
        # the scan: full (the instantiations with the layout AND the scan's length compiled in) or — half of the cases, round 6 — of any
        # shorter length (the ones with the layout alone), on a filter whose REQUESTED measurement capacity may be smaller than the
        # layout's (phd_create rounds it up)
        m_scan = M if rng.random() < 0.5 else int(rng.integers(1, M + 1))
        mm_req = M if rng.random() < 0.5 else int(rng.integers(m_scan, M + 1))
        w = S.make_workload(N, G, M, seed=seed, clustered=clustered)
        over = dict(n_particles=N)
        if ft:
            over.update(filterType=1, maxCardinality=255)
        cfg = P.default_config(**over)
        dz = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
        dn = torch.from_numpy(w["noise"][0].copy()).to(dev)
        out = []
        fast = False
        try:
            for general in ("1", "0"):
                os.environ["PHD_LAYOUT"] = general
                with P.PhdFilter(cfg, n_particles=N, map_capacity=cap, max_measurements=mm_req) as f, \
                        P.PhdFilter(cfg, n_particles=N, map_capacity=cap, max_measurements=mm_req) as g:
                    for x in (f, g):
                        x.set_particles(w["poses"], w["logw"])
                        x.set_maps(w["maps"], w["sizes"])
                    f.predict((2.0, 0.05), w["noise"][0])
                    f.update(w["z"][0][:m_scan])
                    st = f.status(raise_on_overflow=False)
                    staged = (f.get_maps(), f.weight_increments())
                    torch.cuda.synchronize()
                    g.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), m_scan, float(w["uniform"][0]), force_resample=bool(seed & 1))
                    g.sync()
                    pg, lg = g.get_particles()
                    which = (L.phd_debug_update_instantiation(f._h), L.phd_debug_update_instantiation(g._h))
                    if general == "1":
                        fast = all(k >= 18 for k in which)      # (no fast path where the build and the layout do not meet: layout 2 three per CU, layout 1 two per CU)
                        if fast:
                            assert all((18 <= k < 27) == (m_scan == M) for k in which), (which, m_scan, M)
                    else:
                        assert all(0 <= k < 18 for k in which), which
                    out.append((staged, (g.get_maps(), pg, lg), st))
            (sa, fa, sta), (sb, fb, stb) = out
            if sta["status"] or stb["status"]:
                # more survivors than the filter's capacity: the step reports PHD_ERR_CAPACITY and which survivors the truncated list
                # keeps depends on their arrival order — not a comparison
                assert (sta["status"], sta["max_survivors"]) == (stb["status"], stb["max_survivors"]), "status %s vs %s" % (sta, stb)
                n_skip += 1
                seed += 1
                continue
            assert sta == stb, "status %s vs %s" % (sta, stb)
            assert np.array_equal(sa[1].view(np.uint32), sb[1].view(np.uint32)), "log-weight increments"
            assert np.array_equal(fa[1], fb[1]) and np.array_equal(fa[2].view(np.uint32), fb[2].view(np.uint32)), "fused particles"
            for p in range(N):
                assert sa[0][p].tobytes() == sb[0][p].tobytes(), "staged map of particle %d" % p
                assert fa[0][p].tobytes() == fb[0][p].tobytes(), "fused map of particle %d" % p
            n_ok += 1
            n_fast += int(fast)
        except AssertionError as e:
            n_fail += 1
            print("FAIL seed %d layout=%d ft=%d N=%d G=%d clustered=%s scan %d of %d (requested capacity %d): %s" % (seed, layout, ft, N, G, clustered, m_scan, M, mm_req, str(e)[:200]))
        seed += 1
    os.environ.pop("PHD_LAYOUT", None)
    print("fast-path fuzz: %d cases bit-identical to the general instantiations (%d of them through the fast path), %d failed, %d skipped (capacity), %.0f s, seeds up to %d"
          % (n_ok, n_fast, n_fail, n_skip, time.time() - t0, seed - 1))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
