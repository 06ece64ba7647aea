"""The C++ multi-device host (include/phdslam_multi.h, libphdslam_multi.so) on hardware: one filter sharded over several
shards must equal a single filter BIT FOR BIT — particles, weights, maps — whatever the shard count, the transport (RCCL on a
one-rank communicator; stream-ordered device copies when shards share the one GPU of a test box) and the exchange form
(whole-shard all-gather / all-to-all of the migrants), with forced and nEff-triggered resampling, host noise and the
device generator (which draws by GLOBAL particle index)."""
import importlib

import numpy as np
import pytest

from parity_utils import pkg, synthetic

pytestmark = pytest.mark.gpu


def mod():
    return importlib.import_module("cuda-phdslam_amd.multi")


def run_single(cfg, w, steps, cap, M, device_rng, force_pattern, extra=None):
    P = pkg()
    out = []
    with P.PhdFilter(cfg, n_particles=w["N"], map_capacity=cap, max_measurements=M) as f:
        f.seed(77)
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        for k in range(steps):
            f.predict((2.0, 0.05 - 0.01 * k), None if device_rng else w["noise"][k])
            f.update(w["z"][k])
            if force_pattern[k]:
                f.resample(w["uniform"][k])
                did = True
            else:
                did, _ = f.resample_if_needed(w["uniform"][k], had_measurements=True)
            if extra is not None and extra[k]:
                f.resample(0.5 * float(w["uniform"][k]))            # resampleParticles again, no update in between
            p, lw = f.get_particles()
            out.append((did, p, lw, f.get_maps()))
        f.status()
    return out


@pytest.mark.parametrize("shards,exchange", [(1, "gathered"), (1, "alltoall"), (1, "pull"), (2, "gathered"), (2, "alltoall"),
                                             (2, "pull"), (3, "pull"), (4, "alltoall"), (4, "pull"), (8, "gathered"), (8, "pull")])
@pytest.mark.parametrize("device_rng", [False, True])
def test_sharded_filter_equals_one_filter(shards, exchange, device_rng):
    P, S, MM = pkg(), synthetic(), mod()
    N, G, M, steps = (64, 14, 9, 5) if shards != 3 else (96, 14, 9, 5)
    w = S.make_workload(N, G, M, seed=300 + shards, n_meas_sets=steps)
    # a skewed weight vector so that the nEff trigger fires on some steps and not on others
    w["logw"] = (w["logw"] + np.linspace(0, 3.0, N).astype(np.float32)).astype(np.float32)
    cfg = P.default_config(n_particles=N, resampleThresh=0.6)
    force = [True, False, False, True, False]
    ref = run_single(cfg, w, steps, 96, 16, device_rng, force)
    with MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=96, max_measurements=16,
                        exchange={"gathered": MM.EXCHANGE_GATHERED, "alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL}[exchange]) as m:
        assert m.n_shards == shards and m.n == N
        assert m.uses_rccl == (shards == 1)              # one shard: a one-rank RCCL communicator; more on one GPU: device copies
        assert m.gathered == (exchange == "gathered") and m.exchange == exchange
        m.seed(77)
        m.set_particles(w["poses"], w["logw"])
        m.set_maps(w["maps"], w["sizes"])
        fired = []
        for k in range(steps):
            did = m.step((2.0, 0.05 - 0.01 * k), None if device_rng else w["noise"][k], w["z"][k], w["uniform"][k],
                         force_resample=force[k])
            fired.append(did)
            p, lw = m.get_particles()
            maps = m.get_maps()
            rdid, rp, rlw, rmaps = ref[k]
            assert did == rdid, (k, did, rdid)
            assert np.array_equal(p, rp), k
            assert np.array_equal(lw, rlw), (k, np.abs(lw - rlw).max())
            for a, b in zip(maps, rmaps):
                assert np.array_equal(a, b), k
        assert any(f and not fo for f, fo in zip(fired, force)) or any((not f) for f in fired)   # the trigger was exercised
        e, gmap, who, poses, lw = m.state_snapshot()
        assert m.last_report.status == 0
    # the snapshot against the single filter's state extraction (pose accumulated on the host in double here)
    with P.PhdFilter(cfg, n_particles=N, map_capacity=96, max_measurements=16) as f:
        f.set_particles(ref[-1][1], ref[-1][2])
        f.set_maps(np.zeros((N, 0), P.GAUSSIAN), np.zeros(N, np.int32))
        e1 = f.expected_pose()
    for fld in ("px", "py", "ptheta"):
        assert abs(e[fld] - e1[fld]) < 1e-5
    assert who == int(np.argmax(ref[-1][2])) and np.array_equal(gmap, ref[-1][3][who])


def test_frozen_steps_and_expected_map():
    """the bench protocol on the sharded filter (frozen: every step restarts from the same snapshot), and the EAP map of the
    global set reduced on shard 0 == the single filter's"""
    P, S, MM = pkg(), synthetic(), mod()
    N = 48
    w = S.make_workload(N, 12, 8, seed=17)
    cfg = P.default_config(n_particles=N)
    with MM.MultiFilter(cfg, n_shards=3, devices=[0, 0, 0], map_capacity=64, max_measurements=16) as m, \
            P.PhdFilter(cfg, n_particles=N, map_capacity=64, max_measurements=16) as f:
        for x in (m, f):
            x.set_particles(w["poses"], w["logw"])
            x.set_maps(w["maps"], w["sizes"])
        assert len(m.expected_map()) == len(f.expected_map())
        assert np.array_equal(m.expected_map(), f.expected_map())
        m.set_frozen(True)
        m.upload_inputs(w["noise"][0], w["z"][0])
        for _ in range(3):
            m.step_resident((2.0, 0.05), 0.4, force_resample=True)
        m.sync()
        p, lw = m.get_particles()
        assert np.array_equal(p, w["poses"]) and np.array_equal(lw, w["logw"])
        for a, b in zip(m.get_maps(), [w["maps"][q] for q in range(N)]):
            assert np.array_equal(a, b)
        m.set_frozen(False)
        m.step_resident((2.0, 0.05), 0.4, force_resample=True)
        assert not np.array_equal(m.get_particles()[0], w["poses"])


def test_rejects_what_it_cannot_shard():
    P, MM = pkg(), mod()
    with pytest.raises(P.PhdError):
        MM.MultiFilter(P.default_config(n_particles=50), n_shards=4, devices=[0] * 4)             # 50 % 4 != 0
    with pytest.raises(P.PhdError):                                                               # the shotgun needs the PULL exchange
        MM.MultiFilter(P.default_config(n_particles=64, nPredictParticles=2), n_shards=2, devices=[0, 0], exchange=MM.EXCHANGE_ALLTOALL)
    with pytest.raises(P.PhdError):
        MM.MultiFilter(P.default_config(n_particles=64), n_shards=2, devices=[0, 0], transport=MM.TRANSPORT_RCCL)


def test_bench_gpus_2_unlaunched_runs_the_cpp_host():
    """`python3 bench.py --gpus 2` with no launcher: ONE process drives both shards through
    libphdslam_multi.so.  On this one-GPU box PHD_BENCH_SHARE_GPU=1 puts both shards on device 0 (device-copy transport):
    a dry run of the path the 2/4/8-GPU scaling record takes, configs[3] (16384 x 256 x 64), strong scaling."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    rec = os.path.join(tempfile.mkdtemp(prefix="phd_bench_"), "bench_last.json")
    env = dict(os.environ, PHD_BENCH_SHARE_GPU="1", PHD_BENCH_RECORD=rec)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--preroll-ms", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # the last stdout line is the compact record (round 6); the riders, phases and the first-contact verdict are in the record file
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert len(last) <= 4096 and line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and line["unit"] == "steps/s"
    assert line["config"]["n_shards"] == 2 and line["config"]["N"] == 16384 and len(line["riders_steps_per_s"]) == 1
    d = json.load(open(rec))
    assert d["value"] == pytest.approx(line["value"], rel=1e-5)
    c = d["config"]
    assert c["cpp_multi_host"] and c["n_shards"] == 2 and c["particles_total"] == 16384 and c["share_gpu_dry_run"]
    # the first-contact check: before timing, the sharded filter equalled a single filter bit for bit on these devices
    assert c["multi_gpu_verified"]["equal_to_single_filter"] is True and c["multi_gpu_verified"]["exchange"] == c["multi_gpu_exchange"], c["multi_gpu_verified"]
    ph = c["multi_gpu_phase_us_shard0"]
    for k in ("local_step", "all_gather", "weights", "plan_export", "send_recv", "import", "resample_with_migration"):
        assert ph[k] >= 0.0
    assert ph["local_step"] > 100.0                       # 8192 particles x 256 x 64 on one GPU: hundreds of microseconds
    assert d["roofline"]["kernel_avg_us"] > 0
    one = [s for s in d["secondary"] if s["config"]["particles_total"] == 16384]
    assert len(one) == 1 and one[0]["value"] > 0          # the N = 1 point, measured in the same run
    # more GPUs than the box has, without the dry-run switch: a clear refusal, not a hang
    env.pop("PHD_BENCH_SHARE_GPU")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 8:
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


@pytest.mark.parametrize("shards", [1, 2, 4])
def test_particle_shotgun_on_shards_equals_one_filter(shards):
    """n_predict_particles = 2 on a sharded filter (VERDICT r2 missing #5; src/phdfilter.cu:1185-1238, trigger src/main.cpp:1286):
    every shard's set doubles per predict (children share the parent's slab through the indirection), the log-weight
    all-gather, the global normalisation and the resample run over the GROWN global set, and the PULL exchange brings it back
    to n_particles when 5 n is exceeded (or nEff drops) — particles, weights and maps bit for bit those of a single filter"""
    P, S, MM = pkg(), synthetic(), mod()
    n, k, steps = 48, 2, 5
    w = S.make_workload(n, 12, 8, seed=630 + shards, n_meas_sets=steps)
    cfg = P.default_config(nPredictParticles=k, n_particles=n, resampleThresh=0.0)      # only N > 5 n triggers ...
    rng = np.random.default_rng(5)
    with P.PhdFilter(cfg, n_particles=n, map_capacity=96, max_measurements=16) as f, \
            MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=96, max_measurements=16) as m:
        assert m.exchange == "pull"
        for x in (f, m):
            x.set_particles(w["poses"], w["logw"])
            x.set_maps(w["maps"], w["sizes"])
        counts = []
        for step in range(steps):
            na = f.n
            noise = np.stack([rng.normal(0, 0.03, na * k), rng.normal(0, 1.0, na * k)], 1).astype(np.float32)
            force = step == 1                                                         # ... and one forced resample of a grown set
            f.predict((2.0, 0.05), noise)
            f.update(w["z"][step])
            if force:
                f.resample(w["uniform"][step])
            else:
                f.resample_if_needed(w["uniform"][step], had_measurements=True)
            did = m.step((2.0, 0.05), noise, w["z"][step], w["uniform"][step], force_resample=force)
            counts.append((f.n, m.n_now, did))
            assert f.n == m.n_now, counts
            pa, la = f.get_particles()
            pb, lb = m.get_particles()
            assert np.array_equal(pa, pb) and np.array_equal(la, lb), (step, counts)
            for x, y in zip(f.get_maps(), m.get_maps()):
                assert np.array_equal(x, y), step
        # 48 -> 96 -> (forced) 48 -> 96 -> 192 -> 384 (> 240: back to 48)
        assert [c[0] for c in counts] == [96, 48, 96, 192, 48], counts
        f.status()


@pytest.mark.parametrize("shards", [1, 2, 4])
@pytest.mark.parametrize("style", ["step", "update_resample"])
def test_shotgun_crossing_5n_on_a_step_without_a_scan(shards, style):
    """control-only steps (timestamped data, subdivide_predict) grow a shotgun particle set too, and the
    step that crosses 5 n_particles may carry no scan (src/main.cpp:1286: `... || N > 5 n_particles`).  No update precedes that
    resample, so no log-weight all-gather and no global normalisation has run: the shards' scratch copy of the global weights
    is stale (shorter than the grown set).  The host gathers every shard's CURRENT normalised weights instead and uses them as
    they are — particles, weights and maps bit for bit those of a single filter, through phd_multi_step and through the
    driver's order phd_multi_update + phd_multi_resample."""
    P, S, MM = pkg(), synthetic(), mod()
    n, k, steps = 32, 2, 6
    w = S.make_workload(n, 10, 8, seed=640 + shards, n_meas_sets=steps)
    cfg = P.default_config(nPredictParticles=k, n_particles=n, resampleThresh=0.0)      # only N > 5 n triggers
    rng = np.random.default_rng(6)
    empty = np.zeros(0, P.MEAS)
    scans = [w["z"][0], empty, empty, w["z"][3], empty, empty]                            # 64, 128, 256 (> 160: on an EMPTY scan) ...
    with P.PhdFilter(cfg, n_particles=n, map_capacity=96, max_measurements=16) as f, \
            MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=96, max_measurements=16) as m:
        for x in (f, m):
            x.set_particles(w["poses"], w["logw"])
            x.set_maps(w["maps"], w["sizes"])
        counts = []
        for step in range(steps):
            na = f.n
            z = scans[step]
            noise = np.stack([rng.normal(0, 0.03, na * k), rng.normal(0, 1.0, na * k)], 1).astype(np.float32)
            f.predict((2.0, 0.05), noise)
            if len(z):
                f.update(z)
            did_f, _ = f.resample_if_needed(w["uniform"][step], had_measurements=len(z) > 0)
            if style == "step":
                did = m.step((2.0, 0.05), noise, z, w["uniform"][step], force_resample=False)
            else:
                m.update((2.0, 0.05), noise, z)
                did = m.n_now > 5 * n                                                      # the driver's trigger (compat/phdslam_main.cpp)
                if did:
                    m.resample(w["uniform"][step])
            counts.append((f.n, m.n_now, did_f, did))
            assert did == did_f and f.n == m.n_now, counts
            pa, la = f.get_particles()
            pb, lb = m.get_particles()
            assert np.array_equal(pa, pb) and np.array_equal(la, lb), (step, counts)
            for x, y in zip(f.get_maps(), m.get_maps()):
                assert np.array_equal(x, y), step
        assert [c[0] for c in counts] == [64, 128, 32, 64, 128, 32], counts
        assert counts[2][2] and counts[5][2]                                               # both resamples fell on empty scans
        f.status()


@pytest.mark.parametrize("shards,exchange", [(2, "pull"), (3, "alltoall"), (2, "gathered")])
def test_resample_without_a_preceding_update_equals_one_filter(shards, exchange):
    """resampleParticles twice in a row, and after a control-only step, on a sharded filter: the second call finds no fresh
    global normalisation and must gather the current weights (phd_global_resample_launch_normalized) — a single filter's
    phd_resample uses its weights as they are, and so must this (a re-normalisation would move last bits)"""
    P, S, MM = pkg(), synthetic(), mod()
    N = 48
    w = S.make_workload(N, 10, 6, seed=77, n_meas_sets=2)
    w["logw"] = (w["logw"] + np.linspace(0, 4.0, N).astype(np.float32)).astype(np.float32)
    cfg = P.default_config(n_particles=N)
    ex = {"gathered": MM.EXCHANGE_GATHERED, "alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL}[exchange]
    with P.PhdFilter(cfg, n_particles=N, map_capacity=64, max_measurements=16) as f, \
            MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=64, max_measurements=16, exchange=ex) as m:
        for x in (f, m):
            x.set_particles(w["poses"], w["logw"])
            x.set_maps(w["maps"], w["sizes"])

        def same(tag):
            pa, la = f.get_particles()
            pb, lb = m.get_particles()
            assert np.array_equal(pa, pb) and np.array_equal(la, lb), tag
            for x, y in zip(f.get_maps(), m.get_maps()):
                assert np.array_equal(x, y), tag
        f.resample(0.31); m.resample(0.31); same("resample of the initial weights (no update at all)")
        f.predict((2.0, 0.05), w["noise"][0]); f.update(w["z"][0])
        m.update((2.0, 0.05), w["noise"][0], w["z"][0]); same("update")
        f.resample(0.62); m.resample(0.62); same("resample after the update")
        f.resample(0.17); m.resample(0.17); same("a second resample, no update in between")
        f.predict((2.0, 0.02), w["noise"][1])
        m.update((2.0, 0.02), w["noise"][1], np.zeros(0, P.MEAS)); same("control-only step")
        f.resample(0.88); m.resample(0.88); same("resample after a control-only step")


def test_auto_exchange_without_peer_access_falls_back_to_alltoall(monkeypatch):
    """a PHD_EXCHANGE_AUTO create on devices WITHOUT peer access must take the host-planned exchange
    (index download + send/recv pairs) without an error.  No such machine is at hand, so the decision is tested with the
    injected PHD_MULTI_FLAG_NO_PEER_ACCESS: AUTO -> alltoall (and still bit for bit a single filter), an explicit PULL -> a
    clear refusal, small shards -> gathered as before; PHD_MULTI_EXCHANGE overrides what AUTO picks, never an explicit option."""
    P, S, MM = pkg(), synthetic(), mod()
    N, steps = 64, 3
    w = S.make_workload(N, 12, 8, seed=91, n_meas_sets=steps)
    cfg = P.default_config(n_particles=N)
    big = dict(n_shards=2, devices=[0, 0], map_capacity=96, max_measurements=16, gathered_limit_bytes=1)   # never "small shards"
    with MM.MultiFilter(cfg, flags=MM.FLAG_NO_PEER_ACCESS, **big) as m:
        assert m.exchange == "alltoall"
        ref = run_single(cfg, w, steps, 96, 16, False, [True] * steps)
        m.seed(77)
        m.set_particles(w["poses"], w["logw"])
        m.set_maps(w["maps"], w["sizes"])
        for k in range(steps):
            m.step((2.0, 0.05 - 0.01 * k), w["noise"][k], w["z"][k], w["uniform"][k], force_resample=True)
            p, lw = m.get_particles()
            assert np.array_equal(p, ref[k][1]) and np.array_equal(lw, ref[k][2]), k
            for a, b in zip(m.get_maps(), ref[k][3]):
                assert np.array_equal(a, b), k
    with MM.MultiFilter(cfg, **big) as m:
        assert m.exchange == "pull"                                   # with peer access (shards on one device always have it)
    with pytest.raises(P.PhdError) as e:
        MM.MultiFilter(cfg, flags=MM.FLAG_NO_PEER_ACCESS, exchange=MM.EXCHANGE_PULL, **big)
    assert "peer access" in str(e.value)
    with MM.MultiFilter(cfg, n_shards=2, devices=[0, 0], map_capacity=96, max_measurements=16, flags=MM.FLAG_NO_PEER_ACCESS) as m:
        assert m.exchange == "gathered"                               # small shards never needed peer access
    monkeypatch.setenv("PHD_MULTI_EXCHANGE", "alltoall")
    with MM.MultiFilter(cfg, **big) as m:
        assert m.exchange == "alltoall"
    with MM.MultiFilter(cfg, exchange=MM.EXCHANGE_PULL, **big) as m:
        assert m.exchange == "pull"                                   # an explicit option wins
    monkeypatch.setenv("PHD_MULTI_EXCHANGE", "bogus")
    with pytest.raises(P.PhdError):
        MM.MultiFilter(cfg, **big)


def test_phase_timing_holds_the_longest_step():
    """a non-forced step that resamples through the all-to-all exchange records 8 phase marks (start, local
    step, all-gather, weights of the normalise, weights of the index launch, plan + export, send/recv, import); the timing
    pass used to hold 7 and dropped the last span silently"""
    P, S, MM = pkg(), synthetic(), mod()
    N = 64
    w = S.make_workload(N, 12, 8, seed=92)
    w["logw"] = (w["logw"] + np.linspace(0, 6.0, N).astype(np.float32)).astype(np.float32)     # nEff well below the threshold
    cfg = P.default_config(n_particles=N, resampleThresh=0.9)
    with MM.MultiFilter(cfg, n_shards=2, devices=[0, 0], map_capacity=96, max_measurements=16, exchange=MM.EXCHANGE_ALLTOALL) as m:
        m.set_particles(w["poses"], w["logw"])
        m.set_maps(w["maps"], w["sizes"])
        m.set_frozen(True)
        m.timing(True)
        m.upload_inputs(w["noise"][0], w["z"][0])
        for _ in range(3):
            m.step_resident((2.0, 0.05), 0.4, force_resample=False)
        ph, k = m.timing_read()
        assert k == 3 and ph["import"] > 0 and ph["send_recv"] > 0 and ph["plan_export"] > 0 and ph["weights"] > 0, ph


@pytest.mark.parametrize("exchange", ["alltoall", "pull"])
def test_phase_timing_survives_a_run_of_staged_calls(exchange):
    """phd_multi_update / phd_multi_resample record phase marks too, and only phd_multi_step_resident used to
    empty the list — with timing on, the sixth staged call or so failed THE FILTER STEP with "more phase marks … than the
    timing pass holds".  Each staged call now opens and closes its own list: 12 update + resample pairs run, every pair counts
    as one step, and the sharded filter still equals a single one."""
    P, S, MM = pkg(), synthetic(), mod()
    N, steps = 64, 12
    w = S.make_workload(N, 12, 8, seed=93, n_meas_sets=steps)
    cfg = P.default_config(n_particles=N)
    ex = {"alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL}[exchange]
    ref = run_single(cfg, w, steps, 96, 16, False, [True] * steps)
    with MM.MultiFilter(cfg, n_shards=2, devices=[0, 0], map_capacity=96, max_measurements=16, exchange=ex, gathered_limit_bytes=1) as m:
        m.seed(77)
        m.set_particles(w["poses"], w["logw"])
        m.set_maps(w["maps"], w["sizes"])
        m.timing(True)
        for k in range(steps):
            m.update((2.0, 0.05 - 0.01 * k), w["noise"][k], w["z"][k])
            m.resample(w["uniform"][k])
            p, lw = m.get_particles()
            assert np.array_equal(p, ref[k][1]) and np.array_equal(lw, ref[k][2]), k
        ph, n = m.timing_read()
        assert n == steps and ph["local_step"] > 0 and ph["all_gather"] > 0 and ph["weights"] > 0 and ph["import"] > 0, (n, ph)


@pytest.mark.parametrize("shards,exchange", [(2, "pull"), (4, "alltoall"), (1, "pull")])
def test_sharded_filter_above_4096_particles_equals_one_filter(shards, exchange):
    """8192 particles: every shard runs the BLOCK FORM of the weights routine (round 5: several workgroups, two grid-wide barriers,
    the bits a function of n alone) on the gathered vector, the single filter runs it fused behind its update workgroups (grid
    N + W) and staged — particles, weights and maps bit for bit, forced and nEff-triggered resamples"""
    P, S, MM = pkg(), synthetic(), mod()
    N, steps = 8192, 3
    w = S.make_workload(N, 6, 4, seed=411, n_meas_sets=steps)
    w["logw"] = (w["logw"] + np.linspace(0, 5.0, N).astype(np.float32)).astype(np.float32)
    cfg = P.default_config(n_particles=N, resampleThresh=0.6)
    ex = {"alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL}[exchange]
    force = [True, False, True]
    ref = run_single(cfg, w, steps, 32, 8, False, force)
    with MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=32, max_measurements=8, exchange=ex, gathered_limit_bytes=1) as m:
        m.seed(77)
        m.set_particles(w["poses"], w["logw"])
        m.set_maps(w["maps"], w["sizes"])
        for k in range(steps):
            did = m.step((2.0, 0.05 - 0.01 * k), w["noise"][k], w["z"][k], w["uniform"][k], force_resample=force[k])
            assert did == ref[k][0], k
            p, lw = m.get_particles()
            assert np.array_equal(p, ref[k][1]) and np.array_equal(lw, ref[k][2]), k
            for a, b in zip(m.get_maps(), ref[k][3]):
                assert np.array_equal(a, b), k


def _copy_free_count(m):
    return [pkg()._lib.lib().phd_debug_copy_free_resamples(m.shard_handle(k)) for k in range(m.n_shards)]


@pytest.mark.parametrize("shards,exchange", [(1, "pull"), (2, "pull"), (4, "pull"), (2, "alltoall"), (4, "alltoall")])
@pytest.mark.parametrize("filter_type", [0, 1])
def test_copy_free_resample_of_a_shard_equals_one_filter(shards, exchange, filter_type, monkeypatch):
    """A shard's global resample moves no local map (round 5): local parents stay by indirection, a remote parent is parked in
    a guest slab by the first slot it fills and named by the others.  Heavy late weights make most slots of the early shards
    remote; the CPHD variant's cardinality rows follow the same index.  Must equal a single filter bit for bit step after step
    (every update reads through the indirection: local slabs, guests), take the copy-free form on every shard on every step
    that resamples after an update, and equal the copying form (PHD_COPY_FREE=0), which a resample WITHOUT an update in
    between falls back to (its indirection may still name guests)."""
    P, S, MM = pkg(), synthetic(), mod()
    N, steps = 96, 5
    w = S.make_workload(N, 12, 8, seed=1200 + shards, n_meas_sets=steps)
    w["logw"] = (w["logw"] + np.linspace(0, 6.0, N).astype(np.float32)).astype(np.float32)
    kw = dict(n_particles=N, resampleThresh=0.5)
    if filter_type:
        kw.update(filterType=1, maxCardinality=63)
    cfg = P.default_config(**kw)
    ex = {"alltoall": MM.EXCHANGE_ALLTOALL, "pull": MM.EXCHANGE_PULL}[exchange]

    def run(copy_free):
        monkeypatch.setenv("PHD_COPY_FREE", "1" if copy_free else "0")
        out = []
        with P.PhdFilter(cfg, n_particles=N, map_capacity=96, max_measurements=16) as f, \
                MM.MultiFilter(cfg, n_shards=shards, devices=[0] * shards, map_capacity=96, max_measurements=16, exchange=ex) as m:
            for x in (f, m):
                x.set_particles(w["poses"], w["logw"])
                x.set_maps(w["maps"], w["sizes"])
            n_free = 0
            for s in range(steps):
                f.predict((2.0, 0.05), w["noise"][s]); f.update(w["z"][s]); f.resample(w["uniform"][s])
                m.step((2.0, 0.05), w["noise"][s], w["z"][s], w["uniform"][s], force_resample=True)
                n_free += 1
                if s == 2:
                    # resampleParticles again, nothing in between: the shards' indirection still names guests
                    f.resample(0.37)
                    m.resample(0.37)
                if s != 1:                                   # (one step goes on without the host looking at the maps)
                    pa, la = f.get_particles()
                    pb, lb = m.get_particles()
                    assert np.array_equal(pa, pb) and np.array_equal(la, lb), (copy_free, s)
                    ma, mb = f.get_maps(), m.get_maps()
                    for x, y in zip(ma, mb):
                        assert np.array_equal(x, y), (copy_free, s)
                    out.append((pb, lb, mb))
            assert _copy_free_count(m) == [n_free if copy_free else 0] * shards
            m.state_snapshot()
            assert m.last_report.status == 0
        return out

    a, b = run(True), run(False)
    for (pa, la, ma), (pb, lb, mb) in zip(a, b):
        assert np.array_equal(pa, pb) and np.array_equal(la, lb)
        for x, y in zip(ma, mb):
            assert np.array_equal(x, y)
