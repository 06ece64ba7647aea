#!/usr/bin/env python3
"""Per-phase time of the fused update+merge kernel from in-kernel stamps (diagnostic build of the
kernel: shares, not absolute run time).  usage: python tools/phase_profile.py [config ids...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("cuda-phdslam_amd"); S = importlib.import_module("cuda-phdslam_amd.synthetic")
NAMES = ["classify+ekf", "pass1 normalisers", "nondetect emit", "pass2 detect emit", "finalise+births", "sort1+permute",
         "merge rounds", "sums A (w, w m)", "cluster means", "sums B (cov) + divide", "append+tail"]
for cid in [int(a) for a in sys.argv[1:]] or [2, 3]:
    w = S.config_workload(cid)
    N, G, M = w["N"], w["G"], w["M"]
    cfg = P.default_config()
    if cid == 5:                                              # the CPHD variant (BASELINE.json configs[4])
        cfg.filterType = 1
        cfg.maxCardinality = 255
    with P.PhdFilter(cfg, n_particles=N, map_capacity=2 * G, max_measurements=M) as f:
        f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"])
        f.set_frozen(True)
        f.debug(2)
        for _ in range(3):
            f.update(w["z"][0])
        import torch
        dz = torch.from_numpy(w["z"][0].view(np.uint8).copy()).cuda(); dn = torch.from_numpy(w["noise"][0].copy()).cuda()
        torch.cuda.synchronize()
        f.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), M, 0.37, True)
        f.sync()
        st = f.stamps().astype(np.int64)
        ws = f.weight_stamps.astype(np.int64)
        print("   weights kernel phases (us): load+normalise %.2f, neff %.2f, det_exp %.2f, scan %.2f, guard %.2f, search+commit %.2f"
              % tuple(np.diff(ws[:7]) * 0.01))
        if ws[7]:
            print("      inside search+commit: index searches %.2f us, copy_particles of the poses / parents + weights %.2f us"
                  % ((ws[7] - ws[5]) * 0.01, (ws[6] - ws[7]) * 0.01))
        d = np.diff(st[:, :12], axis=1) * 0.01  # us
        tot = (st[:, 11] - st[:, 0]) * 0.01
        span = (st[:, 11].max() - st[:, 0].min()) * 0.01
        print("config %d (%dx%dx%d): per-workgroup mean %.1f us, max %.1f us; kernel span %.1f us; status %s" %
              (cid, N, G, M, tot.mean(), tot.max(), span, f.status()))
        for k, name in enumerate(NAMES):
            print("   %-20s mean %8.2f us  (%5.1f %%)   max %8.2f" % (name, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean(), d[:, k].max()))
        if st[:, 23].any():
            print("   inside 'pass1 normalisers' (thread 0's wave): the (feature x measurement) sweep + candidate list %.2f us, wait for the "
                  "slowest wave %.2f us, log Z + births + particle weight %.2f us" %
                  (((st[:, 23] - st[:, 1]) * 0.01).mean(), ((st[:, 24] - st[:, 23]) * 0.01).mean(), ((st[:, 2] - st[:, 24]) * 0.01).mean()))
        r = st[:, 12:16].astype(np.float64)
        if cid == 5:
            print("   CPHD block (inside 'pass1 normalisers'): staging + birth cardinality %.2f us, forward ESF sweep (wave 0) beside predicted "
                  "cardinality + n-sums (waves 1-7) %.2f us, backward sweep + inner products %.2f us, weights + cardinality update %.2f us"
                  % tuple(r[:, k].mean() * 0.01 for k in range(4)))
            if st[:, 16].any():
                # stamps of other waves (100 MHz), relative to the start of the concurrent phase / of the backward sweep
                g = [((st[:, k] - st[:, 29]) * 0.01).mean() for k in range(16, 21)] + [((st[:, k] - st[:, 30]) * 0.01).mean() for k in (21, 22)]
                print("   (us after the start of the concurrent phase) forward sweep done %.2f | predicted cardinality %.2f, sync %.2f, B_n %.2f, "
                      "n-sums done %.2f | (us after the start of the backward phase) wave 0 done %.2f, wave 3 done %.2f" % tuple(g))
            continue
        print("   rounds: %.1f per particle; matrix %.2f us, resolve %.2f us, assign %.2f us (sums over rounds)" %
              (r[:, 3].mean(), r[:, 0].mean() * 0.01, r[:, 1].mean() * 0.01, r[:, 2].mean() * 0.01))
        if st[:, 31].any():
            print("   one-shot finish of the rounds (merge_tail, inside 'merge rounds'): %.2f us per particle; %.0f %% of the particles took it"
                  % (st[:, 31].mean() * 0.01, 100.0 * (st[:, 31] > 0).mean()))
        if st[:, 28].any():
            c = st[:, 25:29].astype(np.float64).mean(axis=0)
            print("   assign statistics per particle: %.0f (survivor, round) tests; filter-positive seeds per test %.2f; exact decisions per "
                  "test %.2f; hits %.0f" % (c[3], c[0] / c[3], c[1] / c[3], c[2]))
        fz = st[:, 16:23].astype(np.float64).mean(axis=0) * 0.01
        if fz.sum() > 0:
            print("   inside the rounds (us, sums over rounds, thread 0's wave): window copy %.2f | matrix filter %.2f, exact + barrier %.2f | "
                  "seed records + filter %.2f, exact decisions %.2f, wait for the slowest wave %.2f, compaction %.2f" % tuple(fz))
