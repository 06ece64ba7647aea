"""Shared pieces of the end-to-end tests on the reference's bundled simulation
(matlab/simData2_ackerman.mat, re-exported to tests/golden/sim_ackerman_e2e.npz by make_golden.py):
331 scans of a range-bearing sensor (about 7 true detections + 20 clutter points per scan, range noise
1 m, bearing noise 2 degrees, 10 m range, 360 degree field of view), the true Ackerman trajectory, the
noise-free controls (dt = 1 s) and the 50 landmarks with the step at which each was first seen.

Sensor parameters are the generator's (matlab/SynthSetup2.m:29-34; the noise levels are confirmed by
the residuals of measurementsTrue against the landmark positions); filter parameters are those of the
reference's cfg/config.cfg."""
import os

import numpy as np
from scipy.optimize import linear_sum_assignment

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SENSOR = dict(maxRange=10.0, maxBearing=3.141593, stdRange=1.0, stdBearing=0.0349, pd=0.95)
CLUTTER_RATE = 20.0
VEHICLE = dict(l=2.83, h=0.76, a=3.78, b=0.5)        # python/AckermanMotionModel.py parameters of the data set
ODOMETRY = dict(stdEncoder=0.2, stdAlpha=0.03)


def load():
    d = np.load(os.path.join(GOLD, "sim_ackerman_e2e.npz"))
    seen = np.unpackbits(d["seen"], axis=1)[:, :len(d["landmarks"])].astype(bool)
    scans = [d["z"][d["z_offsets"][k]:d["z_offsets"][k + 1]] for k in range(len(d["z_offsets"]) - 1)]
    return dict(traj=d["traj"], u=d["u"], dt=d["dt"], scans=scans, landmarks=d["landmarks"], seen=seen)


def clutter_density():
    f32 = np.float32
    return float(f32(CLUTTER_RATE) / (f32(2) * f32(SENSOR["maxBearing"]) * f32(SENSOR["maxRange"])))   # src/main.cpp:1065-1066


def ospa(X, Y, c=5.0, p=1.0):
    """python/ospa.py:220-274 with scipy's assignment solver"""
    X, Y = np.asarray(X, float).reshape(-1, 2), np.asarray(Y, float).reshape(-1, 2)
    m, n = len(X), len(Y)
    if m == 0 and n == 0:
        return 0.0
    if m == 0 or n == 0:
        return c
    if m > n:
        X, Y, m, n = Y, X, n, m
    D = np.minimum(np.hypot(X[:, None, 0] - Y[None, :, 0], X[:, None, 1] - Y[None, :, 1]), c)
    r, cc = linear_sum_assignment(D)
    return float((((D[r, cc] ** p).sum() + c ** p * (n - m)) / n) ** (1.0 / p))


def confirmed(gmap, threshold=0.5):
    return gmap[gmap["weight"] > threshold]["mean"].astype(np.float64)


def oracle_config(O, **over):
    kw = dict(SENSOR, clutterDensity=clutter_density(), dt=1.0, **VEHICLE)
    kw.update(over)
    return O.default_config(**kw)


def scan_struct(dtype, scan):
    z = np.zeros(len(scan), dtype)
    z["range"], z["bearing"] = scan[:, 0], scan[:, 1]
    return z


def oracle_mapping(O, data, n_steps=None):
    """BASELINE.json configs[0]: one particle on the true trajectory (follow_trajectory, src/main.cpp:1239-1243)"""
    cfg = oracle_config(O)
    gmap = np.zeros(0, O.GAUSSIAN)
    for k, scan in enumerate(data["scans"][:n_steps]):
        pose = np.zeros(1, O.POSE)
        pose["px"], pose["py"], pose["ptheta"] = data["traj"][k]
        gmap = O.update_particle(pose[0], gmap, scan_struct(O.MEAS, scan), cfg)["map"]
    return gmap
