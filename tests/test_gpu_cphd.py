"""SURVEY.md §8(f) N4 / BASELINE.json configs[4] on hardware: the CPHD variant (filter_type = 1) through the
C-ABI against the CPU oracle (oracle/cphd_cpu.c — parity unpinned at the reference boundary, see its
header; the oracle itself is checked against an arbitrary-precision evaluation in test_cphd_oracle.py).

Tolerances (fp32 log-domain recursion on both sides, expf/log1pf of glibc vs the ROCm device library):
the PHD path's value tolerances (tests/parity_utils.py), particle log-weight increment 2e-3 + (M + 2) ulps, log cardinality
4e-3 on the entries above -40 — each <= 10 x the maxima observed (printed under -s).  The merge stage is, as for the PHD path,
bit-exact on the GPU's own survivors, and every particle is compared cluster by cluster under the device's decisions."""
import importlib
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O
from parity_utils import (OBS, assert_maps_close, compare_particle_with_oracle, oracle_config_from, oracle_full_cphd_update, pkg,
                          synthetic, ulp32)

pytestmark = pytest.mark.gpu

CN_ATOL = 4e-3                   # log cardinality rows (entries above -40); observed 4.1e-4 (profiles/r04_parity_observed.txt)


def cphd_dlogw_tol(ref, M):
    """CPHD log-weight increment log<Y0,p>: a log-sum-exp over max_cardinality + 1 terms on top of the M-term structure;
    observed 8.5e-4 over tools/fuzz_cphd.py (|increment| ~ 270), 6.1e-5 at 4096 x 256 x 64"""
    return 2e-3 + (M + 2) * ulp32(ref)


@pytest.fixture(autouse=True)
def _print_observed(request):
    OBS.clear()
    yield
    line = OBS.report(request.node.name)
    if line:
        print("\n" + line)


def log_poisson(mean, nmax):
    from math import lgamma
    n = np.arange(nmax + 1)
    return (n * np.log(mean) - mean - np.array([lgamma(k + 1) for k in n])).astype(np.float32)


def random_priors(rng, N, nmax, centre):
    cn = rng.normal(0, 1.0, (N, nmax + 1)) - 0.05 * (np.arange(nmax + 1)[None, :] - centre) ** 2
    cn -= np.log(np.exp(cn).sum(axis=1, keepdims=True))
    return cn.astype(np.float32)


def make(cfg, w, cap=None, mm=64):
    P = pkg()
    f = P.PhdFilter(cfg, n_particles=w["N"], map_capacity=cap or 2 * w["G"], max_measurements=mm)
    f.set_particles(w["poses"], w["logw"])
    f.set_maps(w["maps"], w["sizes"])
    return f


# (the last case: a long cardinality distribution on a small filter — the CPHD block's arrays do not fit the idle survivor planes
#  behind the sweep rows and take the space behind the common LDS layout instead, csrc/phd_cphd.h cphd_block_in_planes)
@pytest.mark.parametrize("N,G,M,nmax,seed", [(12, 24, 10, 63, 1), (8, 48, 33, 255, 2), (6, 16, 70, 127, 3), (6, 8, 8, 1023, 4)])
def test_cphd_update_matches_oracle(N, G, M, nmax, seed, min_structural=0.5):
    """min_structural: the share of particles expected to come out with the ORACLE's own cluster structure (every particle is held
    to the oracle either way: a particle without it has every differing decision proven, parity_utils).  The randomised sweep
    (tools/fuzz_cphd.py) passes 0: with 130 births of nearly equal weight in 4 particles, explained seed-order inversions in three
    of them are not a failure (seed 37043 of the round-6 long sweep: same outcome with and without round 6's merge changes)."""
    P, S = pkg(), synthetic()
    cfg = P.default_config(filterType=1, maxCardinality=nmax)
    ocfg = oracle_config_from(cfg)
    w = S.make_workload(N, G, M, seed=seed)
    rng = np.random.default_rng(seed)
    prior = random_priors(rng, N, nmax, G)
    with make(cfg, w, cap=2 * G + M + 8, mm=max(M, 16)) as f:
        assert f.cardinalities().shape == (N, nmax + 1)
        assert np.allclose(f.cardinalities(), -np.log(nmax + 1), atol=1e-6)          # uniform start (src/main.cpp:1142)
        f.set_cardinalities(prior)
        f.debug(True)
        f.update(w["z"][0])
        f.status()
        maps = f.get_maps()
        dlw = f.weight_increments()
        cn = f.cardinalities()
        n_struct = 0
        for p in range(N):
            gmap = w["maps"][p, :w["sizes"][p]]
            ref = oracle_full_cphd_update(w["poses"][p], gmap, w["z"][0], ocfg, cfg.clutterRate, prior[p])
            live = ref["cn"] > -40
            OBS.note("cphd_cardinality_row_abs", np.abs(cn[p][live] - ref["cn"][live]).max())
            assert np.allclose(cn[p][live], ref["cn"][live], atol=CN_ATOL), (p, np.abs(cn[p][live] - ref["cn"][live]).max())
            assert abs(np.log(np.exp(cn[p].astype(np.float64)).sum())) < 2e-3
            # log-weight increment, merge stage bit for bit on the GPU's own survivors, survivor set up to members proven
            # marginal, the map cluster by cluster under the device's decisions with every flip proven (parity_utils)
            surv, sidx = f.survivors(p)
            r = compare_particle_with_oracle(maps[p], surv, sidx, ref, ocfg, M, dlw=dlw[p], what="particle %d" % p,
                                             tail_bit_exact=False, dlogw_tol=cphd_dlogw_tol(ref["dlogw"], M))
            n_struct += bool(r["structural"])
        assert n_struct >= (N // 2 if min_structural >= 0.5 else int(min_structural * N))


def test_cphd_with_poisson_prior_is_the_phd_filter():
    """the algebraic identity the oracle test checks on the CPU, end to end on the device: maps of a CPHD
    filter started from Poisson(<1,map>) cardinalities equal the PHD filter's"""
    P, S = pkg(), synthetic()
    w = S.make_workload(16, 32, 12, seed=5)
    cfg_p = P.default_config()
    cfg_c = P.default_config(filterType=1, maxCardinality=255)
    with make(cfg_p, w) as fp, make(cfg_c, w) as fc:
        prior = np.stack([log_poisson(float(w["maps"][p, :w["sizes"][p]]["weight"].sum()), 255) for p in range(16)])
        fc.set_cardinalities(prior)
        fp.update(w["z"][0])
        fc.update(w["z"][0])
        mp, mc = fp.get_maps(), fc.get_maps()
        for p in range(16):
            assert_maps_close(mc[p], mp[p], w_rtol=4e-3, what="particle %d" % p)
        # particle weights: equal up to the particle-independent constant, i.e. equal after normalisation
        _, lp = fp.get_particles()
        _, lc = fc.get_particles()
        assert np.abs(lp - lc).max() < 2e-2


def test_cphd_cardinalities_follow_the_particles():
    P, S = pkg(), synthetic()
    N = 40
    cfg = P.default_config(filterType=1, maxCardinality=63)
    w = S.make_workload(N, 16, 8, seed=9, n_meas_sets=3)
    with make(cfg, w) as f:
        f.update(w["z"][0])
        before = f.cardinalities()
        maps_before = f.get_maps()
        idx = f.resample(0.42)
        assert len(np.unique(idx)) < N
        after = f.cardinalities()
        assert np.array_equal(after, before[idx])                                  # copy_particles carries them
        est, who = f.cardinality_estimate()
        assert np.array_equal(est, after[who])
        # next update reads each particle's own (inherited) row
        f.predict((2.0, 0.03), None)
        f.update(w["z"][1])
        cn2 = f.cardinalities()
        dup = [j for j in range(1, N) if idx[j] == idx[j - 1]]
        assert dup and not np.array_equal(cn2[dup[0]], after[dup[0]])
        assert np.all(np.abs(np.log(np.exp(cn2.astype(np.float64)).sum(axis=1))) < 2e-3)
        # set_maps on a resampled filter keeps the rows with their particles
        f.resample(0.9)
        cn3 = f.cardinalities()
        m3 = f.get_maps()
        f.set_maps(np.zeros((N, 4), P.GAUSSIAN), np.zeros(N, np.int32))
        assert np.array_equal(f.cardinalities(), cn3) and len(m3) == N and len(maps_before) == N


def _worker(rank, world, port, out_dir, N, G, M, seed, u, exchange):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    D = importlib.import_module("cuda-phdslam_amd.dist")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    w = S.make_workload(N, G, M, seed=seed)
    n = N // world
    sl = slice(rank * n, (rank + 1) * n)
    cfg = P.default_config(n_particles=N, filterType=1, maxCardinality=63)
    f = P.PhdFilter(cfg, n_particles=n, map_capacity=4 * G, max_measurements=M, global_particles=N, global_offset=rank * n)
    f.set_particles(w["poses"][sl], w["logw"][sl])
    f.set_maps(w["maps"][sl], w["sizes"][sl])
    d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
    d_noise = torch.from_numpy(w["noise"][0][sl].copy()).to(dev)
    torch.cuda.synchronize()
    shard = D.GpuShard(f, N)
    sf = D.ShardedFilter(shard, N, rank, world)
    if exchange == "alltoall":
        sf.gathered_limit = 0
    if exchange == "rows":
        # the local step writes maps, poses, counts, raw weights AND the cardinality rows straight into the export rows
        rows = shard.step_local_rows((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M)
        idx = sf.resample_gathered(u, weights_in_rows=True, want_idx=True, rows=rows)
    else:
        f.predict_dev((2.0, 0.05), d_noise.data_ptr())
        shard.update_local_dev(d_z.data_ptr(), M)
        sf.normalize(sf.gather_logweights())
        idx = sf.resample(u)                                  # "gathered": separate export, indices from normalize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), idx=idx, cn=f.cardinalities(), lw=f.get_particles()[1])
    f.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["alltoall", "gathered", "rows"])
def test_cphd_two_ranks_equal_one_filter(tmp_path, exchange):
    """the cardinality rows migrate with the particles (export / import behind the slab), in every form of the exchange"""
    import torch.multiprocessing as mp
    P, S = pkg(), synthetic()
    N, G, M, seed, u, world = 48, 16, 8, 31, 0.27, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), N, G, M, seed, u, exchange), nprocs=world, join=True)
    w = S.make_workload(N, G, M, seed=seed)
    with P.PhdFilter(P.default_config(n_particles=N, filterType=1, maxCardinality=63), n_particles=N, map_capacity=4 * G,
                     max_measurements=M) as f:
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        f.predict((2.0, 0.05), w["noise"][0])
        f.update(w["z"][0])
        idx = f.resample(u)
        cn = f.cardinalities()
    n = N // world
    moved = 0
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(d["idx"], idx)
        assert np.array_equal(d["cn"], cn[r * n:(r + 1) * n])
        moved += int(np.sum(idx[r * n:(r + 1) * n] // n != r))
    assert moved > 0


def test_cphd_config5_size():
    """BASELINE.json configs[4]: 4096 particles with the per-particle cardinality update (256 x 64 per particle)"""
    P, S = pkg(), synthetic()
    w = S.config_workload(5)
    cfg = P.default_config(filterType=1, maxCardinality=255)
    ocfg = oracle_config_from(cfg)
    outs = []
    for rep in range(2):
        with make(cfg, w, mm=w["M"]) as f:
            f.predict((2.0, 0.05), w["noise"][0])
            f.update(w["z"][0])
            f.status()
            cn = f.cardinalities()
            _, lw = f.get_particles()
            maps = f.get_maps()
            outs.append((cn, lw, maps))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])      # deterministic
    cn, lw, maps = outs[0]
    assert np.all(np.isfinite(lw)) and abs(np.exp(lw.astype(np.float64)).sum() - 1) < 1e-4
    assert np.all(np.abs(np.log(np.exp(cn.astype(np.float64)).sum(axis=1))) < 5e-3)
    # posterior mean cardinality tracks the posterior mass of the map
    mean_n = (np.exp(cn.astype(np.float64)) * np.arange(cn.shape[1])).sum(axis=1)
    mass = np.array([float(m["weight"].astype(np.float64).sum()) for m in maps])
    assert np.abs(mean_n - mass).mean() < 0.05 * mass.mean()
    # sampled particles against the oracle
    ref_poses = O.predict_ackerman(w["poses"], 0.05, 2.0, w["noise"][0], ocfg)
    prior = np.full(256, -np.log(256.0), np.float32)
    for p in (0, 1777, 4095):
        ref = O.cphd_update_particle(ref_poses[p], w["maps"][p, :w["sizes"][p]], w["z"][0], ocfg, cfg.clutterRate, prior)
        live = ref["cn"] > -40
        assert np.allclose(cn[p][live], ref["cn"][live], atol=CN_ATOL), np.abs(cn[p][live] - ref["cn"][live]).max()
        assert abs(len(maps[p]) - len(ref["map"])) <= 2


def test_cphd_with_a_spill_list_fused_equals_staged():
    """the CPHD variant on a dense scan whose survivor lists exceed the LDS capacity (M = 200 > 64: the mantissa / exponent ESF
    sweeps; 2 310 survivors: phd_merge_spill_kernel): the single-launch step against the staged calls, bit for bit, over two
    steps with a resample in between (the spill merge reads the parent's slab, the cardinality rows follow the particles)"""
    import torch
    P, S = pkg(), synthetic()
    N, G, M = 6, 300, 200
    w = S.make_workload(N, G, M, seed=4242, clustered=True, n_meas_sets=2)
    cfg = P.default_config(filterType=1, maxCardinality=255, clutterRate=200.0)
    dev = torch.device("cuda:0")

    def mk():
        f = P.PhdFilter(cfg, n_particles=N, map_capacity=768, max_measurements=200, survivor_capacity=4096)
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        return f
    with mk() as a, mk() as b:
        b.debug(4)
        for k in range(2):
            dz = torch.from_numpy(w["z"][k].view(np.uint8).copy()).to(dev)
            dn = torch.from_numpy(w["noise"][k].copy()).to(dev)
            torch.cuda.synchronize()
            a.step_dev((2.0, 0.05), dn.data_ptr(), dz.data_ptr(), M, w["uniform"][k], force_resample=True)
            a.sync()
            b.predict((2.0, 0.05), w["noise"][k])
            b.update(w["z"][k])
            b.resample(w["uniform"][k])
            sa, sb = a.status(), b.status()
            assert sa == sb and sa["status"] == 0 and sa["max_survivors"] > 2048, (sa, sb)
            pa, la = a.get_particles()
            pb, lb = b.get_particles()
            assert np.array_equal(pa, pb) and np.array_equal(la, lb)
            for x, y in zip(a.get_maps(), b.get_maps()):
                assert np.array_equal(x, y)
            assert np.array_equal(a.cardinalities(), b.cardinalities())
