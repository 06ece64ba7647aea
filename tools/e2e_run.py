#!/usr/bin/env python3
"""Run the `phdslam` driver end to end on the reference's bundled simulation (tests/golden/sim_ackerman_e2e.npz)
with N particles and report what a user of the executable sees: wall time per filter step (loopTime.log,
src/main.cpp:1300-1305: predict + update + state extraction + log writing + resample) and the estimation
quality of the last log (pose error, OSPA of the logged map against the simulation's landmarks).

    python tools/e2e_run.py [N=256] [out_dir]
"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import importlib
    from e2e_utils import load
    P = importlib.import_module("cuda-phdslam_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    d = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] not in ("", "-") else tempfile.mkdtemp(prefix="phd_e2e_")
    os.makedirs(d, exist_ok=True)
    data = load()
    with open(os.path.join(d, "measurements.txt"), "w") as f:
        f.write("% range bearing pairs, one scan per line\n")
        for scan in data["scans"]:
            f.write(" ".join("%.6f %.6f" % (r, b) for r, b in scan) + " \n")
    with open(os.path.join(d, "controls.txt"), "w") as f:
        f.write("% velocity\tsteering angle\n")
        for v, a in data["u"]:
            f.write("%.6f %.6f\n" % (v, a))
    cfg = open(os.path.join(ROOT, "tests", "golden", "config_sample.cfg")).read()
    repl = dict(max_range="10.0", std_range="1.0", std_bearing="0.0349", dt="1.0", l="2.83", h="0.76", a="3.78", b="0.5",
                std_encoder="0.2", std_alpha="0.03", n_particles=str(n), data_directory=d + "/",
                initial_x="%.6f" % data["traj"][0, 0], initial_y="%.6f" % data["traj"][0, 1],
                initial_yaw="%.6f" % data["traj"][0, 2])
    for k, v in repl.items():
        cfg, cnt = re.subn(r"^%s\s*=.*$" % k, "%s = %s" % (k, v), cfg, flags=re.M)
        assert cnt == 1, k
    cfg_path = os.path.join(d, "config.cfg")
    open(cfg_path, "w").write(cfg)
    out = os.path.join(d, "logs")
    os.makedirs(out, exist_ok=True)
    for fn in os.listdir(out):
        os.remove(os.path.join(out, fn))
    r = subprocess.run([os.path.join(ROOT, "cuda-phdslam_amd", "bin", "phdslam"), cfg_path, "synth", "--out", out, "--seed", "7",
                        "--capacity", "512"] + sys.argv[3:], capture_output=True, text=True)
    if r.returncode:
        print(r.stdout[-2000:], r.stderr[-2000:])
        raise SystemExit(r.returncode)
    for line in r.stdout.splitlines():
        if line.startswith("loop profile"):               # PHD_DRIVER_PROFILE=1
            print(line)
    t = np.loadtxt(os.path.join(out, "loopTime.log"))
    # which instantiation of the update kernel the bundled scans ran (csrc/phd_kernels.hip, include/phdslam.h: phd_debug_update_instantiation)
    inst = [int(m.group(1)) for m in re.finditer(r"inst=(-?\d+)", r.stdout)]
    ms = [int(m.group(1)) for m in re.finditer(r" M=(\d+) ", r.stdout)]
    if inst:
        hist = {k: inst.count(k) for k in sorted(set(inst))}
        print("update-kernel instantiation over the %d steps: %s (below 18: everything from the arguments; 27 ... 35: the filter's LDS layout compiled in, "
              "%d steps; 18 ... 26: the layout and a full scan compiled in, %d steps); measurements per step min %d, median %d, max %d"
              % (len(inst), hist, sum(v for k, v in hist.items() if k >= 27), sum(v for k, v in hist.items() if 18 <= k < 27), min(ms), int(np.median(ms)), max(ms)))
    pf = os.path.join(out, "loopProfile.log")
    if os.path.exists(pf):                                    # PHD_DRIVER_PROFILE=1: which phase do the slow steps spend their time in?
        q = np.loadtxt(pf)
        names = ["inputs+predict", "update (synchronised for the profile)", "state extraction", "resample", "log hand-off"]
        first = open(pf).readline()
        if first.startswith("# pipelined"):               # the pipelined loop: what the HOST does per iteration
            names = ["wait for the helper thread's draws", "enqueue the step (7 calls)", "wait for the device (previous step)", "log hand-off",
                     "time file + progress line"]
        ph = q[:, 4:9]
        tot = ph.sum(1)
        slow = tot >= np.percentile(tot, 90)
        print("   per-phase ms        median     p90     max | mean over the slowest 10 %% of steps (those above %.3f ms)" % np.percentile(tot, 90))
        for k, nm in enumerate(names):
            print("   %-38s %7.3f %7.3f %7.3f | %7.3f" % (nm, np.median(ph[:, k]), np.percentile(ph[:, k], 90), ph[:, k].max(), ph[slow, k].mean()))
        print("   slow steps: measurements %.1f (all steps %.1f), map size %.0f (%.0f), resampled %.0f %% (%.0f %%); corr(step time, step index) %.2f, "
              "corr(step time, map size) %.2f, corr(step time, measurements) %.2f"
              % (q[slow, 1].mean(), q[:, 1].mean(), q[slow, 2].mean(), q[:, 2].mean(), 100 * q[slow, 3].mean(), 100 * q[:, 3].mean(),
                 np.corrcoef(tot, q[:, 0])[0, 1], np.corrcoef(tot, q[:, 2])[0, 1], np.corrcoef(tot, q[:, 1])[0, 1]))
    L = P._lib.lib()
    res = np.zeros(5, np.float64)
    k = len(data["scans"]) - 1
    tp = np.ascontiguousarray(data["traj"][k, :2], np.float32)
    truth = np.ascontiguousarray(data["landmarks"], np.float32)
    rc = L.phd_evaluate_state_log(os.path.join(out, "state_estimate%05d.log" % k).encode(), tp.ctypes.data_as(C.c_void_p),
                                  truth.ctypes.data_as(C.c_void_p), len(truth), 1.0, 5.0, res.ctypes.data_as(C.c_void_p))
    assert rc == 0
    print("phdslam on the bundled simulation: %d particles, %d steps; loop time mean %.3f ms, median %.3f ms, p90 %.3f ms "
          "(predict + update + state extraction + log + resample, host I/O included); last step: pose error %.2f m, "
          "OSPA %.2f m (localisation %.2f, cardinality %.2f), nEff %.1f"
          % (n, len(t), t.mean(), np.median(t), np.percentile(t, 90), res[0], res[1], res[2], res[3], res[4]))


if __name__ == "__main__":
    main()
