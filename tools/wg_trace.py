#!/usr/bin/env python3
"""Experiment (libraries built with -DPHD_EXP_TRACE): start / hand-off / end of every workgroup of the fused step and the CU it
ran on.  usage: PHD_LIB=.../libphdslam_trace.so python tools/wg_trace.py <config id>"""
import ctypes
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import torch
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    c = S.CONFIGS[cfg_id]
    N, G, M = c["N"], c["G"], c["M"]
    w = S.make_workload(N, G, M, seed=0x5EED0000 + cfg_id, clustered=c["clustered"])
    f, ts = bench.make_filter(P, torch, cfg_id, N, G, M, N, 0, dev, 0)
    f.set_particles(w["poses"], w["logw"])
    f.set_maps(w["maps"], w["sizes"])
    d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
    d_noise = torch.from_numpy(w["noise"][0].copy()).to(dev)
    f.set_frozen(True)
    for _ in range(300):
        f.step_dev((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M, 0.37, force_resample=True)
    f.sync()
    lib = ctypes.CDLL(os.environ["PHD_LIB"])
    t = np.zeros((N + 1, 8), np.uint64)
    rc = lib.phd_exp_trace_read(t.ctypes.data_as(ctypes.c_void_p), N + 1)
    assert rc == 0, rc
    t0 = t[:, 0].min()
    st = (t[:, 0] - t0).astype(np.float64) / 100.0
    en = (t[:, 1] - t0).astype(np.float64) / 100.0
    ho = (t[:N, 3] - t0).astype(np.float64) / 100.0
    hw = t[:, 2]
    p2a = (t[:N, 6] - t0).astype(np.float64) / 100.0      # pass 2 done (thread 0's wave)
    p2 = (t[:N, 4] - t0).astype(np.float64) / 100.0       # ... and everybody's (barrier)
    mg = (t[:N, 5] - t0).astype(np.float64) / 100.0       # merge done
    print("thread 0: hand-off -> pass 2 done %.2f -> barrier %.2f -> merge done %.2f -> end %.2f (mean us per segment)"
          % ((p2a - ho).mean(), (p2 - p2a).mean(), (mg - p2).mean(), (en[:N] - mg).mean()))
    print("particles: start mean %.2f max %.2f | hand-off mean %.2f max %.2f | end mean %.2f max %.2f | duration mean %.2f max %.2f (us)"
          % (st[:N].mean(), st[:N].max(), ho.mean(), ho.max(), en[:N].mean(), en[:N].max(), (en - st)[:N].mean(), (en - st)[:N].max()))
    print("weights workgroup: start %.2f end %.2f" % (st[N], en[N]))
    # HW_ID (hwreg 4) bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID (hwreg 20) bits 3:0
    cu = ((hw & 0xFFFFFFFF) >> 8) & 0xF
    se = ((hw & 0xFFFFFFFF) >> 13) & 0x7
    sh = ((hw & 0xFFFFFFFF) >> 12) & 0x1
    xcc = (hw >> 32) & 0xF
    key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
    uniq, cnt = np.unique(key, return_counts=True)
    print("distinct CUs used: %d; workgroups per CU histogram: %s" % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
    for x in np.unique(xcc[:N]):
        sel = np.where(xcc[:N] == x)[0]
        print("  XCC %d: %3d workgroups, hand-off mean %.2f, end mean %.2f max %.2f | pass 2 %.2f, wait %.2f, merge %.2f, tail %.2f" % (
            x, len(sel), ho[sel].mean(), en[sel].mean(), en[sel].max(), (p2a - ho)[sel].mean(), (p2 - p2a)[sel].mean(),
            (mg - p2)[sel].mean(), (en[:N] - mg)[sel].mean()))
    late = np.argsort(-en)[:6]
    for i in late:
        same = np.where(key == key[i])[0]
        print("  wg %4d: start %.2f end %.2f  CU key %05x shared with %s" % (i, st[i], en[i], key[i], [int(j) for j in same if j != i]))
    f.close()


if __name__ == "__main__":
    main()
