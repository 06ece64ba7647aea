#!/usr/bin/env python3
"""Distribution of the end-to-end SLAM quality over noise seeds (the test bars of tests/test_gpu_e2e.py are statistical):
256 particles on the bundled simulation, OSPA (c = 5 m) of the MAP and EAP maps against the 50 landmarks, final pose error.
usage: python tools/e2e_seed_scan.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from e2e_utils import load, ospa
from parity_utils import pkg
import test_gpu_e2e as T
P = pkg(); data = load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sc, se = [], []
for seed in range(s0, s0 + n):
    r = T._slam_run(P, data, 256, seed)
    a, b = ospa(r["est"], data["landmarks"]), ospa(r["est_eap"], data["landmarks"])
    sc.append(a); se.append(b)
    print("seed %2d: OSPA map %.2f  EAP %.2f  pose err %.2f (worst %.2f)  resampled %d  features %d" % (seed, a, b, r["err"], r["worst"], r["n_resampled"], len(r["est"])))
print("MAP: median %.2f  max %.2f | EAP: median %.2f  max %.2f" % (np.median(sc), max(sc), np.median(se), max(se)))
