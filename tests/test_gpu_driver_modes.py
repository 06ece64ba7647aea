"""The driver's input modes of SURVEY.md §8(f) N1 on hardware: asynchronous time-stamped inputs
(src/main.cpp:1187-1229), subdivide_predict, the particle shotgun with the N > 5 n_particles
trigger (:1286), follow_trajectory (:1239-1243) and HEAD's 7-line log — each compared against
the same loop written in Python over the C-ABI."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

from test_gpu_driver import PKG, ROOT, parse_log

pytestmark = pytest.mark.gpu


def write_data(d, n_meas, n_ctrl, cfg_over, times=None, traj=None, seed=8, M=10):
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    w = S.make_workload(1, 16, M, seed=seed, n_meas_sets=n_meas)
    with open(os.path.join(d, "measurements.txt"), "w") as f:
        f.write("% r b pairs\n")
        for k in range(n_meas):
            z = w["z"][k]
            f.write(" ".join("%.6f %.6f" % (z["range"][i], z["bearing"][i]) for i in range(len(z))) + " \n")
    with open(os.path.join(d, "controls.txt"), "w") as f:
        f.write("% v alpha\n")
        for k in range(n_ctrl):
            f.write("%.5f %.6f\n" % (1.5 + 0.2 * k, 0.04 - 0.015 * k))
    if times is not None:
        np.savetxt(os.path.join(d, "measurement_times.txt"), times[0], fmt="%.3f")
        np.savetxt(os.path.join(d, "control_times.txt"), times[1], fmt="%.3f")
    if traj is not None:
        with open(os.path.join(d, "traj.txt"), "w") as f:
            f.write("% px py ptheta vx vy vtheta\n")
            for q in traj:
                f.write("%.5f %.5f %.5f 0 0 0\n" % tuple(q))
    cfg = open(os.path.join(ROOT, "tests", "golden", "config_sample.cfg")).read()
    cfg = cfg.replace("data_directory = /data/synth_bowtie/", "data_directory = %s/" % d)
    for k, v in cfg_over.items():
        import re
        if re.search(r"^%s\s*=.*$" % k, cfg, flags=re.M):
            cfg = re.sub(r"^%s\s*=.*$" % k, "%s = %s" % (k, v), cfg, flags=re.M)
        else:
            cfg += "%s = %s\n" % (k, v)
    path = os.path.join(d, "config.cfg")
    open(path, "w").write(cfg)
    return path


def python_loop(cfg_path, seed, capacity=256):
    """run_synth's loop (src/main.cpp:1178-1312) over the C-ABI, returning per-step records"""
    P = importlib.import_module("cuda-phdslam_amd")
    rng = C.CDLL(os.path.join(PKG, "libphdfilter_compat.so"))
    rng.randn.restype = C.c_double
    rng.randu01.restype = C.c_double
    rng.phd_compat_seed_rng(C.c_uint64(seed))
    cfg, ddir, _ = P.load_config(cfg_path)
    Z = P.load_measurements(os.path.join(ddir, "measurements.txt"))
    U = P.load_controls(os.path.join(ddir, "controls.txt"))
    mt = np.loadtxt(os.path.join(ddir, "measurement_times.txt"), ndmin=1).astype(np.float32) if os.path.exists(os.path.join(ddir, "measurement_times.txt")) else np.zeros(0, np.float32)
    ct = np.loadtxt(os.path.join(ddir, "control_times.txt"), ndmin=1).astype(np.float32) if len(mt) else np.zeros(0, np.float32)
    traj = None
    if cfg.followTrajectory:
        traj = np.loadtxt(os.path.join(ddir, "traj.txt"), comments="%", ndmin=2)
        cfg.n_particles = 1
        cfg.nPredictParticles = 1
    k = max(1, cfg.nPredictParticles)
    n_steps = len(Z) if not len(mt) else len(mt) + len(ct)
    recs = []
    z_idx = c_idx = 0
    last = cur = np.float32(0)
    control = (0.0, 0.0)
    with P.PhdFilter(cfg, n_particles=cfg.n_particles, map_capacity=capacity, max_measurements=max(len(z) for z in Z)) as f:
        for n in range(n_steps):
            if len(mt):
                if z_idx >= len(mt) or c_idx >= len(ct):
                    break
                meas_first, both = mt[z_idx] < ct[c_idx], mt[z_idx] == ct[c_idx]
                last, cur = cur, ct[c_idx]
                cfg.dt = float(np.float32(cur - last))
                f.set_config(cfg)
                zz = np.zeros(0, P.MEAS)
                if meas_first:
                    zz = Z[z_idx]; z_idx += 1
                elif both:
                    control = (U["v_encoder"][c_idx], U["alpha"][c_idx]); c_idx += 1
                    zz = Z[z_idx]; z_idx += 1
                else:
                    control = (U["v_encoder"][c_idx], U["alpha"][c_idx]); c_idx += 1
            else:
                zz = Z[n]
                if n > 0:
                    control = (U["v_encoder"][n - 1], U["alpha"][n - 1])
            if traj is not None:
                q = np.zeros(1, P.POSE)
                q["px"], q["py"], q["ptheta"] = traj[min(n, len(traj) - 1)][:3]
                f.set_particles(q, np.zeros(1, np.float32))
            elif n > 0:
                for _ in range(max(1, cfg.subdividePredict)):
                    m = f.n * k
                    noise = np.zeros((m, 2), np.float32)
                    for i in range(m):
                        noise[i, 0] = cfg.stdAlpha * rng.randn()
                        noise[i, 1] = cfg.stdEncoder * rng.randn()
                    f.predict(control, noise)
            if len(zz):
                f.update(zz)
            e = f.expected_pose()
            gm, _ = f.map_estimate()
            eap = f.expected_map() if (cfg.mapEstimate & 2) and f.n > 1 else None
            if eap is not None and not (cfg.mapEstimate & 1):
                gm = eap
            poses, lw = f.get_particles()
            cn = f.cardinality_estimate()[0] if cfg.filterType == 1 else None
            did, idx = f.resample_if_needed(rng.randu01(), had_measurements=len(zz) > 0)
            recs.append(dict(pose=e, map=gm, lw=lw, poses=poses, did=did, idx=idx, n=len(lw), M=len(zz), eap=eap, cn=cn))
            f.status()
    return recs


def run_driver(cfg_path, out, seed, extra=()):
    os.makedirs(out, exist_ok=True)
    r = subprocess.run([os.path.join(PKG, "bin", "phdslam"), cfg_path, "synth", "--out", out, "--seed", str(seed),
                        "--capacity", "256"] + list(extra), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def compare(recs, out, log7=False):
    for n, rec in enumerate(recs):
        path = os.path.join(out, "state_estimate%05d.log" % n)
        if log7:
            lines = open(path).read().split("\n")
            pose = np.array(lines[0].split(), float)
            gmap = np.array(lines[1].split(), float).reshape(-1, 7)
            lw = np.array(lines[3].split(), float)
            poses = np.array(lines[4].split(), float).reshape(-1, 6)
            ridx = np.array(lines[5].split(), int)
            want = rec["idx"] if rec["did"] else np.arange(rec["n"])
            assert np.array_equal(ridx[:len(want)], want), n
        else:
            pose, gmap, lw, poses = parse_log(path)
        e = rec["pose"]
        assert np.allclose(pose[:3], [e["px"], e["py"], e["ptheta"]], rtol=2e-5, atol=1e-6), n
        assert len(gmap) == len(rec["map"]) and np.allclose(gmap[:, 0], rec["map"]["weight"], rtol=2e-5), n
        assert len(lw) >= rec["n"] and np.allclose(lw[:rec["n"]], rec["lw"], rtol=2e-5), n
        assert np.allclose(poses[:rec["n"], 0], rec["poses"]["px"], rtol=2e-5, atol=1e-6), n
    assert not os.path.exists(os.path.join(out, "state_estimate%05d.log" % len(recs)))


def test_timestamped_inputs_and_subdivide(tmp_path):
    d = str(tmp_path)
    # measurements at 0, 0.1, 0.25, 0.3, 0.55; controls at 0.1, 0.2, 0.3, 0.4: measurement-only, both, odometry-only steps
    mt = np.array([0.0, 0.1, 0.25, 0.3, 0.55], np.float32)
    ct = np.array([0.1, 0.2, 0.3, 0.4], np.float32)
    cfg_path = write_data(d, 5, 4, dict(n_particles=40, subdivide_predict=2, resample_threshold=0.9), times=(mt, ct))
    recs = python_loop(cfg_path, 4)
    assert {r["M"] > 0 for r in recs} == {True, False}       # steps with and without measurements occurred
    run_driver(cfg_path, os.path.join(d, "o"), 4)
    compare(recs, os.path.join(d, "o"))


def test_shotgun_driver_and_log7(tmp_path):
    d = str(tmp_path)
    cfg_path = write_data(d, 7, 7, dict(n_particles=12, n_predict_particles=2, resample_threshold=0.0))
    recs = python_loop(cfg_path, 5)
    assert max(r["n"] for r in recs) > 5 * 12 and any(r["did"] for r in recs)     # grew past 5 n_particles, then resampled
    run_driver(cfg_path, os.path.join(d, "o"), 5, extra=["--log7"])
    compare(recs, os.path.join(d, "o"), log7=True)


def test_follow_trajectory(tmp_path):
    d = str(tmp_path)
    traj = [(0.2 * k, 0.05 * k, 0.01 * k) for k in range(5)]
    cfg_path = write_data(d, 5, 5, dict(follow_trajectory=1, n_particles=30), traj=traj)
    recs = python_loop(cfg_path, 6)
    assert all(r["n"] == 1 for r in recs)
    assert abs(recs[3]["pose"]["px"] - 0.6) < 1e-6
    run_driver(cfg_path, os.path.join(d, "o"), 6)
    compare(recs, os.path.join(d, "o"))


def test_expected_map_in_the_log(tmp_path):
    """map_estimate = 2: the log's map line is the EAP map (recoverSlamState, src/main.cpp:363-379);
    map_estimate = 3: MAP map in the log, EAP map beside it"""
    d = str(tmp_path)
    cfg_path = write_data(d, 4, 4, dict(n_particles=24, map_estimate=2, resample_threshold=0.5))
    recs = python_loop(cfg_path, 9)
    assert all(r["eap"] is not None for r in recs) and len(recs[-1]["map"]) > 0
    run_driver(cfg_path, os.path.join(d, "o2"), 9)
    compare(recs, os.path.join(d, "o2"))
    for n, rec in enumerate(recs):
        _, gmap, _, _ = parse_log(os.path.join(d, "o2", "state_estimate%05d.log" % n))
        assert np.allclose(gmap[:, 1:3], rec["map"]["mean"], rtol=2e-5, atol=1e-5)
    cfg_path = write_data(d, 4, 4, dict(n_particles=24, map_estimate=3, resample_threshold=0.5))
    recs3 = python_loop(cfg_path, 9)
    run_driver(cfg_path, os.path.join(d, "o3"), 9)
    compare(recs3, os.path.join(d, "o3"))
    for n, rec in enumerate(recs3):
        e = np.array(open(os.path.join(d, "o3", "expected_map%05d.log" % n)).read().split(), float).reshape(-1, 7)
        assert len(e) == len(rec["eap"]) and np.allclose(e[:, 0], rec["eap"]["weight"], rtol=2e-5)
        assert np.array_equal(rec["eap"], recs[n]["eap"])            # same filter run, same EAP map


def test_cphd_driver(tmp_path):
    """filter_type = 1: the driver runs the CPHD variant; the log's last line is cn_estimate (src/main.cpp:944-949)"""
    d = str(tmp_path)
    cfg_path = write_data(d, 5, 5, dict(n_particles=20, filter_type=1, max_cardinality=63, resample_threshold=0.6))
    recs = python_loop(cfg_path, 12)
    assert all(r["cn"] is not None for r in recs)
    run_driver(cfg_path, os.path.join(d, "o"), 12)
    compare(recs, os.path.join(d, "o"))
    for n, rec in enumerate(recs):
        last = open(os.path.join(d, "o", "state_estimate%05d.log" % n)).read().split("\n")[4]
        cn = np.array(last.split(), float)
        assert len(cn) == 64 and np.allclose(cn, rec["cn"], rtol=2e-5, atol=1e-6)
    # the cardinality estimate is a normalised distribution that moved away from the uniform start
    pn = np.exp(recs[-1]["cn"].astype(np.float64))
    assert abs(pn.sum() - 1) < 5e-3 and pn.max() > 0.1


def test_shotgun_on_the_sharded_driver(tmp_path):
    """n_predict_particles = 2 through `phdslam --devices 1 --shards 2` (the C++ multi-device host; VERDICT r2 missing #5): the
    grown set is gathered, normalised and resampled back over both shards (PULL exchange) — the logs of the single-device run,
    map / weights / poses lines character for character, the expected pose (summed on the host in double) to the log's digits"""
    d = str(tmp_path)
    cfg_path = write_data(d, 7, 7, dict(n_particles=12, n_predict_particles=2, resample_threshold=0.0))
    run_driver(cfg_path, os.path.join(d, "one"), 5)
    run_driver(cfg_path, os.path.join(d, "two"), 5, extra=["--devices", "1", "--shards", "2"])
    sizes = []
    for n in range(7):
        a = open(os.path.join(d, "one", "state_estimate%05d.log" % n)).read().split("\n")
        b = open(os.path.join(d, "two", "state_estimate%05d.log" % n)).read().split("\n")
        assert a[1:] == b[1:], n
        assert np.allclose([float(x) for x in a[0].split()], [float(x) for x in b[0].split()], rtol=2e-5, atol=1e-6), n
        sizes.append(len(a[2].split()))
    # the log is written before the resample: 12, 24, 48, 96 (> 5 n_particles: resampled back to 12), 24, 48, 96
    assert sizes == [12, 24, 48, 96, 24, 48, 96], sizes
