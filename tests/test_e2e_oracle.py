"""End-to-end sanity of the CPU oracle on the reference's own simulation (SURVEY.md §8c item 3,
BASELINE.json configs[0]): not a bit-level pin — the reference ships no filter outputs — but the check
that the restated algorithm actually maps and localises on the data the reference was developed on."""
import numpy as np

from e2e_utils import ODOMETRY, confirmed, load, oracle_config, oracle_mapping, ospa, scan_struct
from oracle import oracle as O


def test_mapping_on_the_true_trajectory():
    """1 particle x bundled scans: the confirmed map converges on the landmarks that were in view"""
    data = load()
    gmap = oracle_mapping(O, data)
    est = confirmed(gmap)
    truth = data["landmarks"][data["seen"][-1]]
    assert len(truth) == 50 and 35 <= len(est) <= 50
    # range noise is 1 m per scan and a fifth of the landmarks is seen only a few times: OSPA (c = 5 m, p = 1)
    # of the confirmed map against all 50 landmarks
    assert ospa(est, truth) < 1.5
    # the confirmed features themselves are accurate: every one lies within 1 m of a landmark
    d = np.hypot(est[:, None, 0] - truth[None, :, 0], est[:, None, 1] - truth[None, :, 1]).min(axis=1)
    assert d.max() < 1.0 and d.mean() < 0.35
    # the PHD mass tracks the number of landmarks seen often enough to be confirmed
    assert abs(float(gmap["weight"].sum()) - len(est)) < 6
    # halfway through, the estimate is scored against what had been seen by then
    half = oracle_mapping(O, data, 166)
    assert ospa(confirmed(half), data["landmarks"][data["seen"][165]]) < 2.5


def test_slam_with_noisy_odometry():
    """128 particles, control noise on top of the noise-free controls: pose and map stay on track"""
    data = load()
    N, cap = 128, 512
    cfg = oracle_config(O)
    rng = np.random.default_rng(1)
    poses = np.zeros(N, O.POSE)
    poses["px"], poses["py"], poses["ptheta"] = data["traj"][0]
    logw = np.full(N, -np.log(N), np.float32)
    maps = np.zeros((N, cap), O.GAUSSIAN)
    sizes = np.zeros(N, np.int32)
    worst = 0.0
    for k, scan in enumerate(data["scans"]):
        noise = np.zeros((N, 2), np.float32)
        v = alpha = 0.0
        if k > 0:
            v, alpha = (float(x) for x in data["u"][k - 1])
            noise = np.stack([ODOMETRY["stdAlpha"] * rng.standard_normal(N), ODOMETRY["stdEncoder"] * rng.standard_normal(N)],
                             1).astype(np.float32)
        r = O.step(poses, logw, maps, sizes, cap, alpha, v, noise, scan_struct(O.MEAS, scan), cfg, rng.random(), False,
                   n_threads=4)
        assert r["rc"] == 0
        idx = r["idx"]
        resampled = not np.array_equal(idx, np.arange(N))
        poses, maps, sizes = r["poses"][idx], r["maps"][idx], r["sizes"][idx]
        logw = np.full(N, -np.log(N), np.float32) if resampled else r["logw"]
        w = np.exp(logw.astype(np.float64))
        err = np.hypot((w * poses["px"]).sum() - data["traj"][k, 0], (w * poses["py"]).sum() - data["traj"][k, 1])
        worst = max(worst, err)
    assert err < 1.0 and worst < 2.0
    best = int(np.argmax(logw))
    est = confirmed(maps[best, :sizes[best]])
    assert 33 <= len(est) <= 50 and ospa(est, data["landmarks"]) < 1.8
