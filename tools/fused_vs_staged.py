#!/usr/bin/env python3
"""The bench step (phd_step_dev, forced resample, frozen snapshot) timed twice: fused into one launch, and with the fusion
switched off (update launch + weights launch).  usage: python tools/fused_vs_staged.py <config id> [steps]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    import torch
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    c = S.CONFIGS[cfg_id]
    N, G, M = c["N"], c["G"], c["M"]
    w = S.make_workload(N, G, M, seed=0x5EED0000 + cfg_id, clustered=c["clustered"])
    for label, dbg in (("fused", 0), ("staged", 4), ("fused", 0)):
        f, ts = bench.make_filter(P, torch, cfg_id, N, G, M, N, 0, dev, 0)
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
        d_noise = torch.from_numpy(w["noise"][0].copy()).to(dev)
        f.set_frozen(True)
        if dbg:
            f.debug(dbg)
        for _ in range(200):
            f.step_dev((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M, 0.37, force_resample=True)
        f.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            f.step_dev((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M, 0.37, force_resample=True)
        f.sync()
        dt = time.perf_counter() - t0
        print("%-8s %-22s %9.1f steps/s  %7.2f us/step" % (label, os.path.basename(os.environ.get("PHD_LIB", "product")),
                                                          steps / dt, 1e6 * dt / steps))
        f.close()


if __name__ == "__main__":
    main()
