#!/usr/bin/env python3
"""Throughput of the fused step on a workload of chosen sizes and capacities (experiments on workgroup residency:
LDS per workgroup follows map_capacity / survivor_capacity, registers follow the library given by PHD_LIB).
usage: python tools/occupancy_probe.py N G M map_capacity survivor_capacity [steps] [config id: 3 = PHD, 5 = CPHD]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    N, G, M, cap, S_cap = (int(a) for a in sys.argv[1:6])
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 300
    cid = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    import torch
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    S.CONFIGS[cid] = dict(N=N, G=G, M=M, clustered=True, map_capacity=cap, survivor_capacity=S_cap)
    torch.cuda.set_device(0)
    r = bench.run_single(P, S, torch, cid, steps, 20, 0.0, torch.device("cuda", 0), 0, extras=False)
    print("%-40s %9.1f steps/s  %8.2f us/step  kernel %8.2f us  residency %s  max_survivors %s" % (
        os.path.basename(os.environ.get("PHD_LIB", "product")), r["value"], 1e3 * r["ms_per_step"],
        r["roofline"]["kernel_avg_us"], r["config"].get("update_residency"), r["config"].get("max_survivors")))


if __name__ == "__main__":
    main()
