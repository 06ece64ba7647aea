#!/usr/bin/env python3
"""How much of a launch is ramp / imbalance / tail?  Per-workgroup start and end stamps of the update kernel (stamped build),
the occupancy timeline, and what a longest-first dispatch order would give (list scheduling of the measured durations on the
same number of slots).  usage: python tools/tail_analysis.py [config id = 3]"""
import heapq, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("cuda-phdslam_amd"); S = importlib.import_module("cuda-phdslam_amd.synthetic")
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = S.config_workload(cid)
N, G, M = w["N"], w["G"], w["M"]
cfg = P.default_config()
with P.PhdFilter(cfg, n_particles=N, map_capacity=2 * G, max_measurements=M) as f:
    f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"])
    f.set_frozen(True)
    import torch
    dz = torch.from_numpy(w["z"][0].view(np.uint8).copy()).cuda(); dn = torch.from_numpy(w["noise"][0].copy()).cuda()
    torch.cuda.synchronize()
    for _ in range(6):                                        # production (fused) launches: they build the longest-first order
        f.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), M, 0.37, True)
    f.sync()
    f.debug(2)                                                # the stamped instantiation takes the order they left (PHD_NO_LPT=1: none)
    for _ in range(2):
        f.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), M, 0.37, True)
    f.sync()
    st = f.stamps().astype(np.int64)
t0, t1 = st[:, 0] * 0.01, st[:, 11] * 0.01
base = t0.min()
t0 -= base; t1 -= base
dur = t1 - t0
span = t1.max()
print("config %d: %d workgroups, duration mean %.1f us (min %.1f, max %.1f, std %.1f); span %.1f us" % (cid, N, dur.mean(), dur.min(), dur.max(), dur.std(), span))
# concurrency timeline
ev = sorted([(t, 1) for t in t0] + [(t, -1) for t in t1])
cur, peak, area, last = 0, 0, 0.0, 0.0
for t, d in ev:
    area += cur * (t - last); last = t; cur += d; peak = max(peak, cur)
slots = peak
print("peak concurrency %d workgroups; mean concurrency %.1f; sum of durations / peak = %.1f us (%.1f %% of the span)" %
      (peak, area / span, dur.sum() / peak, 100 * dur.sum() / peak / span))
for frac in (0.5, 0.9, 0.95, 0.99):
    # time at which the concurrency last drops below frac * peak for good
    cur = 0; tlast = 0.0
    for t, d in ev:
        cur += d
        if cur >= frac * peak: tlast = t
    print("   concurrency >= %2.0f %% of the peak until %.1f us (the last %.1f us run below)" % (100 * frac, tlast, span - tlast))
print("first start spread: 90 %% of the first %d workgroups started within %.1f us" % (slots, np.sort(t0)[: slots][int(0.9 * slots)]))
# correlation of duration with the survivor count proxy (rounds, phases)
def simulate(order):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for p in order:
        s = heapq.heappop(h); e = s + dur[p]; end = max(end, e); heapq.heappush(h, e)
    return end
print("list scheduling of the measured durations on %d slots: in index order %.1f us, longest first %.1f us, shortest first %.1f us; "
      "lower bound %.1f us" % (slots, simulate(range(N)), simulate(np.argsort(-dur)), simulate(np.argsort(dur)), max(dur.sum() / slots, dur.max())))
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "tail_cfg%d.npy" % cid), np.stack([t0, t1], 1))
