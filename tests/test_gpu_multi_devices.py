"""The C++ multi-device host on DISTINCT devices: everything tests/test_gpu_multi.py checks with all shards on
device 0 — one sharded filter == a single filter bit for bit — with one device per shard: `ncclCommInitAll` over k > 1 devices,
the grouped per-communicator `ncclAllGather`, `hipDeviceEnablePeerAccess` + `phd_resample_pull_kernel` reading a REMOTE slab over
xGMI, the cross-device `ready` / `done` events, `ncclSend`/`ncclRecv` pairs between different GPUs.

Skipped where fewer than two GPUs are visible — which is every box this build has been run on (the pool hands out one MI355X
per call): UNMEASURED ON HARDWARE.  The file exists so that the first machine with two GPUs runs the matrix before anything else
does; `bench.py --gpus k` verifies the same equality on first contact (config key `multi_gpu_verified`)."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from parity_utils import pkg, synthetic

pytestmark = pytest.mark.gpu


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


NDEV = _device_count()
# PHD_TEST_SHARE_DEVICE=1: a dry run of THIS FILE on a one-GPU box — every shard on device 0 (device-copy transport instead of
# RCCL), so that the test bodies themselves have been executed before a machine with two GPUs meets them.  Not a measurement.
SHARE = os.environ.get("PHD_TEST_SHARE_DEVICE") == "1"
needs_two = pytest.mark.skipif(NDEV < 2 and not SHARE,
                               reason="needs >= 2 GPUs: %d visible (multi-device host unmeasured on hardware)" % NDEV)
KS = sorted({2, min(8, NDEV)}) if NDEV >= 2 else ([2, 4] if SHARE else [2])


def devs(k):
    return [0] * k if (SHARE and NDEV < k) else list(range(k))


def mod():
    return importlib.import_module("cuda-phdslam_amd.multi")


def _exchange(MM, name):
    return {"gathered": MM.EXCHANGE_GATHERED, "alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL, "auto": MM.EXCHANGE_AUTO}[name]


@needs_two
@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("exchange", ["pull", "gathered", "alltoall", "auto"])
@pytest.mark.parametrize("device_rng", [False, True])
def test_sharded_over_distinct_devices_equals_one_filter(k, exchange, device_rng):
    from test_gpu_multi import run_single
    P, S, MM = pkg(), synthetic(), mod()
    N, G, M, steps = 32 * k, 14, 9, 6
    w = S.make_workload(N, G, M, seed=900 + k, n_meas_sets=steps)
    w["logw"] = (w["logw"] + np.linspace(0, 3.0, N).astype(np.float32)).astype(np.float32)     # the nEff trigger fires on some steps
    cfg = P.default_config(n_particles=N, resampleThresh=0.6)
    force = [True, False, False, True, False, True]
    ref = run_single(cfg, w, steps, 96, 16, device_rng, force)
    with MM.MultiFilter(cfg, n_shards=k, devices=devs(k), map_capacity=96, max_measurements=16, exchange=_exchange(MM, exchange),
                        gathered_limit_bytes=(1 if exchange == "auto" else 0)) as m:
        assert m.n_shards == k and (m.uses_rccl or SHARE), "distinct devices must form an RCCL communicator"
        if exchange == "auto":
            assert m.exchange in ("pull", "alltoall")             # peer access decides; never an error
        else:
            assert m.exchange == exchange
        m.seed(77)
        m.set_particles(w["poses"], w["logw"])
        m.set_maps(w["maps"], w["sizes"])
        fired = []
        for s in range(steps):
            did = m.step((2.0, 0.05 - 0.01 * s), None if device_rng else w["noise"][s], w["z"][s], w["uniform"][s], force_resample=force[s])
            fired.append(did)
            p, lw = m.get_particles()
            rdid, rp, rlw, rmaps = ref[s]
            assert did == rdid, (s, did, rdid)
            assert np.array_equal(p, rp), s
            assert np.array_equal(lw, rlw), (s, np.abs(lw - rlw).max())
            for a, b in zip(m.get_maps(), rmaps):
                assert np.array_equal(a, b), s
        assert any(f and not fo for f, fo in zip(fired, force)) or any(not f for f in fired)
        e, gmap, who, poses, lw = m.state_snapshot()
        assert m.last_report.status == 0
        assert who == int(np.argmax(ref[-1][2])) and np.array_equal(gmap, ref[-1][3][who])


@needs_two
@pytest.mark.parametrize("k", KS)
def test_migration_heavy_resample_over_distinct_devices(k):
    """a weight vector concentrated on ONE shard: every other shard's slots are filled from remote parents (the pull kernel's
    phase 0 crosses the link, phase 1 fans out locally; the all-to-all form sends one row per destination) — several steps so
    that the `done` events (the owner must not overwrite a slab a peer is still reading) are exercised"""
    from test_gpu_multi import run_single
    P, S, MM = pkg(), synthetic(), mod()
    N, steps = 64 * k, 4
    w = S.make_workload(N, 16, 9, seed=910 + k, n_meas_sets=steps)
    lw = np.full(N, -30.0, np.float32)
    lw[N - 64:N - 60] = 0.0                                                      # four heavy particles on the LAST shard
    w["logw"] = (lw - np.float32(np.log(np.exp(lw.astype(np.float64)).sum()))).astype(np.float32)
    cfg = P.default_config(n_particles=N)
    ref = run_single(cfg, w, steps, 96, 16, False, [True] * steps)
    for exchange in ("pull", "alltoall"):
        with MM.MultiFilter(cfg, n_shards=k, devices=devs(k), map_capacity=96, max_measurements=16, exchange=_exchange(MM, exchange),
                            gathered_limit_bytes=1) as m:
            m.seed(77)
            m.set_particles(w["poses"], w["logw"])
            m.set_maps(w["maps"], w["sizes"])
            for s in range(steps):
                m.step((2.0, 0.05 - 0.01 * s), w["noise"][s], w["z"][s], w["uniform"][s], force_resample=True)
                p, lw2 = m.get_particles()
                assert np.array_equal(p, ref[s][1]) and np.array_equal(lw2, ref[s][2]), (exchange, s)
                for a, b in zip(m.get_maps(), ref[s][3]):
                    assert np.array_equal(a, b), (exchange, s)


@needs_two
@pytest.mark.parametrize("k", KS)
def test_particle_shotgun_over_distinct_devices(k):
    """n_predict_particles = 2 sharded over k devices (PULL brings the grown set back), including a 5 n crossing on an EMPTY scan"""
    P, S, MM = pkg(), synthetic(), mod()
    n, kp, steps = 24 * k, 2, 6
    w = S.make_workload(n, 10, 8, seed=920 + k, n_meas_sets=steps)
    cfg = P.default_config(nPredictParticles=kp, n_particles=n, resampleThresh=0.0)
    rng = np.random.default_rng(8)
    empty = np.zeros(0, P.MEAS)
    scans = [w["z"][0], empty, empty, w["z"][3], w["z"][4], w["z"][5]]
    with P.PhdFilter(cfg, n_particles=n, map_capacity=96, max_measurements=16) as f, \
            MM.MultiFilter(cfg, n_shards=k, devices=devs(k), map_capacity=96, max_measurements=16) as m:
        assert m.exchange == "pull" and (m.uses_rccl or SHARE)
        for x in (f, m):
            x.set_particles(w["poses"], w["logw"])
            x.set_maps(w["maps"], w["sizes"])
        for step in range(steps):
            na = f.n
            noise = np.stack([rng.normal(0, 0.03, na * kp), rng.normal(0, 1.0, na * kp)], 1).astype(np.float32)
            z = scans[step]
            f.predict((2.0, 0.05), noise)
            if len(z):
                f.update(z)
            did_f, _ = f.resample_if_needed(w["uniform"][step], had_measurements=len(z) > 0)
            did = m.step((2.0, 0.05), noise, z, w["uniform"][step], force_resample=False)
            assert did == did_f and f.n == m.n_now, step
            pa, la = f.get_particles()
            pb, lb = m.get_particles()
            assert np.array_equal(pa, pb) and np.array_equal(la, lb), step
            for x, y in zip(f.get_maps(), m.get_maps()):
                assert np.array_equal(x, y), step


@needs_two
@pytest.mark.parametrize("k", KS)
def test_cphd_rows_migrate_over_distinct_devices(k):
    """the CPHD variant's cardinality rows travel with their particles (pull, all-to-all, gathered): every later update reads the
    row its particle inherited, so maps and weights stay bit for bit a single filter's only if the rows migrated"""
    P, S, MM = pkg(), synthetic(), mod()
    N, steps = 24 * k, 3
    w = S.make_workload(N, 12, 8, seed=930 + k, n_meas_sets=steps)
    w["logw"] = (w["logw"] + np.linspace(0, 5.0, N).astype(np.float32)).astype(np.float32)
    cfg = P.default_config(n_particles=N, filterType=1, maxCardinality=63)
    for exchange in ("pull", "alltoall", "gathered"):
        with P.PhdFilter(cfg, n_particles=N, map_capacity=96, max_measurements=16) as f, \
                MM.MultiFilter(cfg, n_shards=k, devices=devs(k), map_capacity=96, max_measurements=16,
                               exchange=_exchange(MM, exchange)) as m:
            for x in (f, m):
                x.set_particles(w["poses"], w["logw"])
                x.set_maps(w["maps"], w["sizes"])
            for s in range(steps):
                f.predict((2.0, 0.05), w["noise"][s]); f.update(w["z"][s]); f.resample(w["uniform"][s])
                m.step((2.0, 0.05), w["noise"][s], w["z"][s], w["uniform"][s], force_resample=True)
                pa, la = f.get_particles()
                pb, lb = m.get_particles()
                assert np.array_equal(pa, pb) and np.array_equal(la, lb), (exchange, s)
                for x, y in zip(f.get_maps(), m.get_maps()):
                    assert np.array_equal(x, y), (exchange, s)


@needs_two
def test_frozen_bench_protocol_and_expected_map_over_distinct_devices():
    P, S, MM = pkg(), synthetic(), mod()
    k, N = 2, 64
    w = S.make_workload(N, 12, 8, seed=940)
    cfg = P.default_config(n_particles=N)
    with MM.MultiFilter(cfg, n_shards=k, devices=devs(2), map_capacity=64, max_measurements=16, gathered_limit_bytes=1) as m, \
            P.PhdFilter(cfg, n_particles=N, map_capacity=64, max_measurements=16) as f:
        for x in (m, f):
            x.set_particles(w["poses"], w["logw"])
            x.set_maps(w["maps"], w["sizes"])
        assert np.array_equal(m.expected_map(), f.expected_map())          # hipMemcpyPeerAsync of the planes to shard 0
        m.set_frozen(True)
        m.timing(True)
        m.upload_inputs(w["noise"][0], w["z"][0])
        for _ in range(4):
            m.step_resident((2.0, 0.05), 0.4, force_resample=True)
        m.sync()
        p, lw = m.get_particles()
        assert np.array_equal(p, w["poses"]) and np.array_equal(lw, w["logw"])
        ph, n = m.timing_read()
        assert n == 4 and ph["local_step"] > 0 and ph["all_gather"] > 0


@needs_two
@pytest.mark.parametrize("k", KS)
def test_bench_gpus_k_unlaunched(k):
    """`python3 bench.py --gpus k --steps 5` with no launcher on k REAL devices: the C++ host, RCCL ranks == k, the first-contact
    verification against a single filter passed, every phase timed"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PHD_BENCH_SHARE_GPU"):
        env.pop(v, None)
    if SHARE and NDEV < k:
        env["PHD_BENCH_SHARE_GPU"] = "1"
    import tempfile
    rec = os.path.join(tempfile.mkdtemp(prefix="phd_bench_"), "bench_last.json")
    env["PHD_BENCH_RECORD"] = rec
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(k), "--steps", "5", "--warmup", "2"], env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)                                   # the compact record; the full one is in the record file
    assert len(last) <= 4096 and line["n_gpus"] == k and line["value"] > 0 and line["config"]["n_shards"] == k
    d = json.load(open(rec))
    c = d["config"]
    assert d["n_gpus"] == k and d["value"] > 0 and c["cpp_multi_host"]
    assert (c["rccl_ranks"] == k and not c.get("share_gpu_dry_run")) or SHARE
    assert c["multi_gpu_verified"]["equal_to_single_filter"] is True, c["multi_gpu_verified"]
    for ph in ("local_step", "all_gather", "weights", "import"):
        assert c["multi_gpu_phase_us_shard0"][ph] >= 0.0


def test_this_file_is_skipped_on_one_gpu_boxes_and_says_so():
    """documentation in executable form: on the one-GPU boxes of this pool every test above is skipped"""
    assert (NDEV >= 2) or SHARE or needs_two.args[0]
