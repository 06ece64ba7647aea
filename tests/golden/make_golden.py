#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own executable artefacts.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python3 tests/golden/make_golden.py

What is pinned (SURVEY.md §8c — everything the reference can execute for this path):
  * ackerman_kat.npz   — sim.control -> sim.traj of matlab/simData2_ackerman.mat (params
                         l=2.83,h=0.76,a=3.78,b=0.5,dt=1) and outputs of the reference's
                         python/AckermanMotionModel.py compute_motion on seeded random inputs.
  * rb_model_kat.npz   — python/RangeBearingMeasurementModel.py compute_measurement /
                         check_in_range / invert_measurement on seeded random inputs.
  * meas_ackerman_head.txt / .npz — first steps of sim.data(k).measurements re-exported in the
                         bundled text format, with the parsed values (loader KAT); checked here
                         against matlab/measurements_synth_ackerman.txt.
Only data (inputs and expected outputs) is written; no reference source is copied.
"""
import os
import sys
import numpy as np
import scipy.io as sio

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, os.path.join(REF, "python"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "python"))
    import AckermanMotionModel as amm
    import RangeBearingMeasurementModel as rbm
    os.chdir(cwd)

    sim = sio.loadmat(os.path.join(REF, "matlab/simData2_ackerman.mat"), squeeze_me=True,
                      struct_as_record=False)["sim"]
    traj = np.asarray(sim.traj, dtype=np.float64)            # 3 x 331
    u = np.stack([np.asarray(c.u, dtype=np.float64) for c in sim.control])  # 330 x 2 (v, alpha)
    dts = np.array([float(c.dt) for c in sim.control])
    params = dict(l=2.83, h=0.76, a=3.78, b=0.5, std_encoder=0.0, std_alpha=0.0)
    model = amm.AckermanMotionModel(params)
    # known-answer: the reference model reproduces sim.traj from sim.control
    pose = traj[:, 0].copy()
    err = 0.0
    for k in range(u.shape[0]):
        pose = model.compute_motion(pose, u[k, 0], u[k, 1], dts[k]).ravel()
        err = max(err, np.abs(pose - traj[:, k + 1]).max())
        pose = traj[:, k + 1].copy()
    print("ackerman KAT: reference python model vs sim.traj, max one-step err = %.3e" % err)
    assert err < 1e-12

    rng = np.random.default_rng(20260101)
    n = 256
    poses = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-np.pi, np.pi, n)], 1)
    ctrl = np.stack([rng.uniform(-3, 5, n), rng.uniform(-0.4, 0.4, n)], 1)
    dt_r = rng.uniform(0.02, 1.0, n)
    params2 = dict(l=1.415, h=0.38, a=1.89, b=0.5, std_encoder=1.0, std_alpha=0.034907)  # cfg/config.cfg:67-72
    model2 = amm.AckermanMotionModel(params2)
    out = np.stack([model2.compute_motion(poses[i], ctrl[i, 0], ctrl[i, 1], dt_r[i]).ravel() for i in range(n)])
    np.savez(os.path.join(OUT, "ackerman_kat.npz"),
             sim_traj=traj, sim_u=u, sim_dt=dts, sim_params=np.array([2.83, 0.76, 3.78, 0.5]),
             rnd_pose=poses, rnd_ctrl=ctrl, rnd_dt=dt_r, rnd_params=np.array([1.415, 0.38, 1.89, 0.5]),
             rnd_out=out)

    # range-bearing model
    sp = dict(max_range=15.0, max_bearing=np.pi, std_range=0.25, std_bearing=0.008727, pd=0.95, clutter_rate=20.0)
    mm = rbm.RangeBearingMeasurementModel(sp)
    n = 512
    pose = np.array([1.5, -2.0, 0.7])
    feats = rng.uniform(-20, 20, (2, n))
    in_range = mm.check_in_range(pose, feats)
    z_all = mm.compute_measurement(pose, feats)               # only the in-range columns
    # narrow FoV variant exercises the bearing test
    sp2 = dict(sp, max_bearing=1.0, max_range=12.0)
    mm2 = rbm.RangeBearingMeasurementModel(sp2)
    in_range2 = mm2.check_in_range(pose, feats)
    z2 = mm2.compute_measurement(pose, feats)
    zz = np.stack([rng.uniform(0.1, 15, n), rng.uniform(-np.pi, np.pi, n)])
    inv = mm.invert_measurement(pose, zz)
    np.savez(os.path.join(OUT, "rb_model_kat.npz"), pose=pose, feats=feats, in_range=in_range, z=z_all,
             max_range2=12.0, max_bearing2=1.0, in_range2=in_range2, z2=z2, zz=zz, inv=inv)

    # loader KAT: sim.data(k).measurements == line k+1 of the bundled text file
    with open(os.path.join(REF, "matlab/measurements_synth_ackerman.txt")) as f:
        lines = f.read().split("\n")
    n_head = 12
    vals, sizes = [], []
    with open(os.path.join(OUT, "meas_ackerman_head.txt"), "w") as f:
        f.write("% measurements re-exported from sim.data(k).measurements: range bearing pairs, one step per line\n")
        for k in range(n_head):
            mk = np.atleast_2d(np.asarray(sim.data[k].measurements, dtype=np.float64))
            if mk.shape[0] != 2:
                mk = mk.reshape(2, -1)
            txt = np.array(lines[k + 1].split(), dtype=np.float64).reshape(-1, 2)
            assert txt.shape[0] == mk.shape[1], (k, txt.shape, mk.shape)
            assert np.abs(txt - mk.T).max() < 1e-6
            f.write(" ".join("%.6f %.6f" % (mk[0, i], mk[1, i]) for i in range(mk.shape[1])) + " \n")
            vals.append(np.array(["%.6f" % v for v in mk.T.ravel()], dtype=np.float64).reshape(-1, 2))
            sizes.append(mk.shape[1])
    n_steps_full = len([l for l in lines[1:] if l.strip()])
    np.savez(os.path.join(OUT, "meas_ackerman_head.npz"), values=np.concatenate(vals), sizes=np.array(sizes),
             n_steps_full=n_steps_full)
    print("loader KAT: %d head steps, bundled file has %d steps" % (n_head, n_steps_full))

    # end-to-end data (BASELINE.json configs[0]): the whole bundled simulation —
    # true trajectory, noise-free controls, every scan (true detections + clutter), the landmarks seen so far
    flat, offs = [], [0]
    for d in sim.data:
        mk = np.atleast_2d(np.asarray(d.measurements, dtype=np.float64))
        if mk.shape[0] != 2:
            mk = mk.reshape(2, -1)
        flat.append(mk.T)
        offs.append(offs[-1] + mk.shape[1])
    gt_final = np.asarray(sim.groundTruth[-1].loc, dtype=np.float64).reshape(2, -1).T
    # groundTruth(k).loc lists the landmarks seen up to step k: every column is one of the final list's
    seen = np.zeros((len(sim.groundTruth), len(gt_final)), bool)
    for k, g in enumerate(sim.groundTruth):
        loc = np.asarray(g.loc, dtype=np.float64).reshape(2, -1).T
        for p in loc:
            j = int(np.argmin(np.abs(gt_final - p).sum(axis=1)))
            assert np.array_equal(gt_final[j], p)
            seen[k, j] = True
        assert seen[k].sum() == len(loc)
    np.savez_compressed(os.path.join(OUT, "sim_ackerman_e2e.npz"), traj=traj.T.astype(np.float32), u=u.astype(np.float32),
                        dt=dts.astype(np.float32), z=np.concatenate(flat).astype(np.float32),
                        z_offsets=np.array(offs, np.int32), landmarks=gt_final.astype(np.float32),
                        seen=np.packbits(seen, axis=1), vehicle=np.array([2.83, 0.76, 3.78, 0.5], np.float32))
    print("e2e data: %d steps, %d measurements, %d landmarks" % (len(sim.data), offs[-1], len(gt_final)))


if __name__ == "__main__":
    main()
