"""Deterministic synthetic range-bearing workloads.

The reference ships python/generate_simdata.py for this job (un-runnable here: needs
`tables`/cPickle); this is the generator bench.py and the parity tests share.  Parameter values
are those of the reference's cfg/config.cfg:46-159.  numpy's PCG64 stream is platform
independent, so a (config id, seed) pair names the same inputs everywhere.
"""
import numpy as np

GAUSSIAN = np.dtype([("cov", np.float32, 4), ("mean", np.float32, 2), ("weight", np.float32)])
POSE = np.dtype([("px", np.float32), ("py", np.float32), ("ptheta", np.float32),
                 ("vx", np.float32), ("vy", np.float32), ("vtheta", np.float32)])
MEAS = np.dtype([("range", np.float32), ("bearing", np.float32), ("label", np.int32)])

# BASELINE.json configs: id -> (particles, Gaussians/particle, measurements/step, clustered landmarks)
CONFIGS = {
    1: dict(N=1, G=64, M=32, clustered=False),
    2: dict(N=256, G=64, M=32, clustered=False),
    3: dict(N=4096, G=256, M=64, clustered=True),
    4: dict(N=16384, G=256, M=64, clustered=True),
    5: dict(N=4096, G=256, M=64, clustered=True),   # CPHD variant (filter_type = 1, max_cardinality 255)
    # not a BASELINE.json config: the densest scan the reference accepts (M clamped to 256, src/phdfilter.cu:3390-3394) on the
    # merge-stress map — ~2 700 survivors per particle, past the 2 048 the LDS merge holds: every particle takes the spill path
    6: dict(N=4096, G=256, M=256, clustered=True, map_capacity=768, survivor_capacity=4096),
}

SENSOR = dict(max_range=15.0, max_bearing=3.141593, std_range=0.25, std_bearing=0.008727)
CONTROL = dict(v=2.0, alpha=0.05, std_encoder=1.0, std_alpha=0.034907)


def make_workload(N, G, M, seed=0x5EED0002, clustered=False, n_meas_sets=1):
    """-> dict(poses[N] POSE, logw[N] f32, maps[N,G] GAUSSIAN, sizes[N] i32, z[n_meas_sets,M] MEAS,
               noise[n_meas_sets,N,2] f32 (n_alpha,n_encoder), uniform[n_meas_sets] f64, truth[G,2])"""
    rng = np.random.default_rng(seed)
    R = SENSOR["max_range"]
    if clustered and G >= 8:
        n_cl = max(G // 8, 1)
        rr = np.sqrt(rng.uniform((1.0 / (0.9 * R)) ** 2, 1.0, n_cl)) * 0.9 * R
        th = rng.uniform(-np.pi, np.pi, n_cl)
        centers = np.stack([rr * np.cos(th), rr * np.sin(th)], 1)
        offs = np.array([[(i % 4) * 0.2, (i // 4) * 0.2] for i in range(8)])
        truth = (centers[:, None, :] + offs[None, :, :]).reshape(-1, 2)[:G]
        if len(truth) < G:
            truth = np.concatenate([truth, truth[:G - len(truth)] + 0.37])
    else:
        rr = np.sqrt(rng.uniform((1.0 / (0.9 * R)) ** 2, 1.0, G)) * 0.9 * R
        th = rng.uniform(-np.pi, np.pi, G)
        truth = np.stack([rr * np.cos(th), rr * np.sin(th)], 1)

    poses = np.zeros(N, POSE)
    poses["px"] = rng.normal(0, 0.05, N)
    poses["py"] = rng.normal(0, 0.05, N)
    poses["ptheta"] = rng.normal(0, 0.01, N)
    lw = -np.log(N) + rng.normal(0, 0.5, N)
    lw = lw - (lw.max() + np.log(np.exp(lw - lw.max()).sum()))
    logw = lw.astype(np.float32)

    maps = np.zeros((N, G), GAUSSIAN)
    mean = truth[None, :, :] + rng.normal(0, 0.1, (N, G, 2))
    s1 = rng.uniform(0.05, 0.5, (N, G)) ** 2
    s2 = rng.uniform(0.05, 0.5, (N, G)) ** 2
    phi = rng.uniform(0, np.pi, (N, G))
    c, s = np.cos(phi), np.sin(phi)
    pxx = c * c * s1 + s * s * s2
    pxy = c * s * (s1 - s2)
    pyy = s * s * s1 + c * c * s2
    maps["mean"] = mean
    maps["cov"][..., 0] = pxx
    maps["cov"][..., 1] = pxy
    maps["cov"][..., 2] = maps["cov"][..., 1]  # exactly symmetric in float32
    maps["cov"][..., 3] = pyy
    maps["weight"] = rng.uniform(0.2, 1.0, (N, G))
    sizes = np.full(N, G, np.int32)

    z = np.zeros((n_meas_sets, M), MEAS)
    n_det = int(0.6 * M)
    for k in range(n_meas_sets):
        pick = rng.integers(0, G, n_det)
        r = np.hypot(truth[pick, 0], truth[pick, 1]) + rng.normal(0, SENSOR["std_range"], n_det)
        b = np.arctan2(truth[pick, 1], truth[pick, 0]) + rng.normal(0, SENSOR["std_bearing"], n_det)
        b = (b + np.pi) % (2 * np.pi) - np.pi
        z["range"][k, :n_det] = np.maximum(r, 0.05)
        z["bearing"][k, :n_det] = b
        z["range"][k, n_det:] = rng.uniform(0.05, R, M - n_det)
        z["bearing"][k, n_det:] = rng.uniform(-SENSOR["max_bearing"], SENSOR["max_bearing"], M - n_det)
    noise = np.zeros((n_meas_sets, N, 2), np.float32)
    noise[..., 0] = rng.normal(0, CONTROL["std_alpha"], (n_meas_sets, N))
    noise[..., 1] = rng.normal(0, CONTROL["std_encoder"], (n_meas_sets, N))
    uniform = rng.uniform(0, 1, n_meas_sets)
    return dict(poses=poses, logw=logw, maps=maps, sizes=sizes, z=z, noise=noise, uniform=uniform,
                truth=truth, N=N, G=G, M=M)


def config_workload(cfg_id, n_particles=None, **kw):
    c = CONFIGS[cfg_id]
    N = c["N"] if n_particles is None else n_particles
    return make_workload(N, c["G"], c["M"], seed=0x5EED0000 + cfg_id, clustered=c["clustered"], **kw)


def shard_workload(w, world, rank):
    """rank's contiguous slice [rank n, (rank + 1) n), n = N / world, of ONE generated workload — the split of BASELINE.json
    configs[3] (bench.py --gpus N): per-particle arrays are sliced, the measurement sets, the resampling uniforms and the
    truth are the same on every rank.  The log-weights stay those of the global set (they sum to one over all ranks)."""
    N = w["N"]
    if N % world:
        raise ValueError("particle count must be divisible by the world size")
    n = N // world
    off = rank * n
    out = dict(w)
    for k in ("poses", "logw", "maps", "sizes"):
        out[k] = w[k][off:off + n]
    out["noise"] = w["noise"][:, off:off + n]
    out["N"] = n
    out["N_global"] = N
    out["offset"] = off
    return out


def algorithmic_bytes(N, G, M, K=None):
    """B_step of SURVEY.md §8(d): 28 B per Gaussian, U = G(M+1)+M update components per particle."""
    K = G if K is None else K
    U = G * (M + 1) + M
    return N * (28 * G + 24 + 2 * 28 * U + 28 * K + 8) + 12 * M + 8
