import importlib, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
P = importlib.import_module("cuda-phdslam_amd"); S = importlib.import_module("cuda-phdslam_amd.synthetic")
for N in (256, 4096):
    G, M = 256, 256
    w = S.make_workload(N, G, M, seed=0x5EED0006, clustered=True)
    cfg = P.default_config(n_particles=N)
    dev = torch.device("cuda:0")
    with P.PhdFilter(cfg, n_particles=N, map_capacity=768, max_measurements=256, survivor_capacity=4096) as f:
        f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"])
        dz = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev); dn = torch.from_numpy(w["noise"][0].copy()).to(dev)
        torch.cuda.synchronize()
        f.set_frozen(True)
        for k in range(3):
            t0 = time.perf_counter()
            f.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), M, 0.3, force_resample=True)
            f.sync()
            print(N, "step", k, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), f.status())
        f.timing_reset(); f.timing(True)
        f.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), M, 0.3, force_resample=True); f.sync()
        print(f.timing_read())
