#!/usr/bin/env python3
"""bench.py — filter-update steps/sec of the GM-PHD-SLAM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2|3]

One "step" = one full filter step on one batch of synthetic input, inputs resident in HBM:
predict -> update (in-range split, births, EKF, PHD weights) -> prune -> merge -> weight
normalise -> nEff -> resample (forced every step).  Steady-state protocol: the
filter is frozen, so every timed iteration restarts from the same device-resident snapshot and
does identical work.

N = 1: BASELINE.json configs[1] (256 particles x 64 Gaussians x 32 measurements).
N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling — every rank holds a
256-particle shard of one 256*N-particle filter; per step one RCCL all-gather of the
un-normalised log-weights, the identical global normalise/resample on every rank and the
migration of particles whose parent lives on another rank.  `value` counts shard-steps: N ranks x
K steps / time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the fused update+prune+merge
kernel, algorithmic bytes of SURVEY.md §8d per launch / its average duration from HIP events on
the filter's stream) and `cpu_baseline` (the CPU oracle timed on the host cores on the same
workload, rank 0 at N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(w, cfg_id, budget_s):
    """the CPU oracle (a port of the reference algorithm: kind "port") on the host cores"""
    from oracle import oracle as O
    N, G, M = w["N"], w["G"], w["M"]
    cap = 2 * G
    ocfg = O.default_config()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # bound the sample: the full particle set at config 2, a slice of it at the big configs
    n = N if N <= 512 else 256
    maps = np.zeros((n, cap), O.GAUSSIAN)
    maps[:, :G] = w["maps"][:n]
    lw = O.normalize_weights(w["logw"][:n])

    cn0 = np.full((n, 256), -np.log(256.0), np.float32)      # config 5: uniform cardinality rows (src/main.cpp:1142)

    def one(threads):
        if cfg_id == 5:
            return O.cphd_step(w["poses"][:n], lw, maps, w["sizes"][:n], cap, 0.05, 2.0, w["noise"][0][:n], w["z"][0], ocfg,
                               20.0, cn0, w["uniform"][0], True, n_threads=threads)
        return O.step(w["poses"][:n], lw, maps, w["sizes"][:n], cap, 0.05, 2.0, w["noise"][0][:n], w["z"][0], ocfg,
                      w["uniform"][0], True, n_threads=threads)

    # the visible core count can exceed what the container may actually use (CPU quota): pick the
    # thread count that is fastest on this host and report THAT as `cores`
    best_t, best_rate = 1, 0.0
    t = 1
    while t <= avail:
        one(t)
        t0 = time.perf_counter()
        one(t)
        rate = 1.0 / (time.perf_counter() - t0)
        if rate > best_rate:
            best_t, best_rate = t, rate
        t *= 2
    t0 = time.perf_counter()
    k = 0
    while True:
        one(best_t)
        k += 1
        el = time.perf_counter() - t0
        if el > budget_s or (k >= 5 and el > 0.5 * budget_s):
            break
    steps_per_s = k / el * (n / N)  # a slice of n particles is n/N of a step
    return {"value": steps_per_s, "unit": "steps/s", "cores": best_t, "kind": "port",
            "sample": "%d steps of the oracle (oracle/scphd_cpu.c + cphd_cpu.c, -O3 -march=native, OpenMP over particles, %d threads "
                      "— the fastest of 1..%d visible) on %d of the %d particles of config %d (%dx%dx%d), %.1f s"
                      % (k, best_t, avail, n, N, cfg_id, N, G, M, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 2000 timed steps of 24 us are 50 ms — long enough for the clocks to settle (400-step runs read ~2 % lower)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=2, help="workload: 2, 3 (BASELINE.json configs[1], [2]) or 5 (configs[4], CPHD)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    D = importlib.import_module("cuda-phdslam_amd.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # PHD_BENCH_SHARE_GPU=1: dry run of the N > 1 path on a one-GPU box (every rank on device 0, gloo transport
    # staged through host memory) — exercises this file's multi-rank code, its numbers are not a measurement
    share = os.environ.get("PHD_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    # PHD_BENCH_ONE_RANK_RCCL=1 (with --gpus 1): the N > 1 step — local step, RCCL all-gather, global resample, RCCL
    # all-to-all — on a ONE-rank RCCL group: what the collective path costs per step before any link is involved.
    # A diagnostic (labelled in config), not the N = 1 measurement.
    one_rank = world == 1 and os.environ.get("PHD_BENCH_ONE_RANK_RCCL") == "1"
    multi = world > 1 or one_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        if share:
            dist.init_process_group("gloo")
        elif one_rank:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    c = S.CONFIGS[args.config]
    N, G, M = c["N"], c["G"], c["M"]
    # every rank's shard: the same distribution, a different seed
    w = S.make_workload(N, G, M, seed=0x5EED0000 + args.config + 1000 * rank, clustered=c["clustered"])
    cfg = P.default_config(n_particles=N * world)
    if args.config == 5:                                      # BASELINE.json configs[4]: the CPHD variant
        cfg.filterType = 1
        cfg.maxCardinality = 255
    # one stream for everything: the filter's kernels and (N > 1) the RCCL collectives torch enqueues
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    f = P.PhdFilter(cfg, n_particles=N, map_capacity=2 * G, max_measurements=M, device=local_rank, stream=stream,
                    global_particles=N * world, global_offset=N * rank)
    lw = w["logw"] - np.float32(np.log(world)) if world > 1 else w["logw"]  # the global set sums to one
    f.set_particles(w["poses"], lw.astype(np.float32))
    f.set_maps(w["maps"], w["sizes"])
    d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
    d_noise = torch.from_numpy(w["noise"][0].copy()).to(dev)
    control = (2.0, 0.05)
    # the resampling uniform is one draw shared by all ranks (every rank must compute the same global indices)
    u = float(np.random.default_rng(0x5EED0000 + args.config).random())
    torch.cuda.synchronize()

    f.set_frozen(True)  # steady state: no step commits, every iteration restarts from the same snapshot
    if not multi:
        def step():
            f.step_dev(control, d_noise.data_ptr(), d_z.data_ptr(), M, u, force_resample=True)
    else:
        shard = D.GpuShard(f, N * world)
        sf = D.ShardedFilter(shard, N * world, rank, world)
        sf.collectives = True
        if os.environ.get("PHD_BENCH_EXCHANGE") == "alltoall":
            sf.gathered_limit = 0
        gathered = sf.gathered()                    # small shards (this config up to 8 ranks): whole-shard all-gather

        def step():
            if gathered:
                # one launch for predict + update + prune + merge, written straight into the export rows (raw weights
                # in the row headers) -> ONE fixed-size RCCL all-gather of the shards -> global normalise + resample
                # indices + import of this shard's parents: nothing waits for the host
                rows = shard.step_local_rows(control, d_noise.data_ptr(), d_z.data_ptr(), M)
                sf.resample_gathered(u, weights_in_rows=True, want_idx=False, rows=rows)
            else:
                # one launch for predict + update + prune + merge + raw weights, then
                shard.step_local_dev(control, d_noise.data_ptr(), d_z.data_ptr(), M)
                # RCCL all-gather of the raw weights, one launch for the global normalise + resample indices (indices
                # to the host for the plan), RCCL all-to-all of the migrating particles, commit
                allw = sf.gather_logweights()
                sf.resample(u, all_raw_logw=allw)   # the bench forces the resample: no host round trip for nEff

    def sync():
        f.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    # HIP events on the stream the kernels run on (torch's current stream IS the filter's stream), bracketing
    # the timed region: GPU time of the K steps without the per-launch event pairs of the pass below
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(ts)
    chunk = max(1, args.steps // 20)            # an (asynchronous) event every `chunk` steps: the spread of the step time
    marks = []
    for k in range(args.steps):
        step()
        if (k + 1) % chunk == 0 and k + 1 < args.steps:
            e = torch.cuda.Event(enable_timing=True)
            e.record(ts)
            marks.append(e)
    ev1.record(ts)
    sync()
    elapsed = time.perf_counter() - t0
    gpu_region_ms = ev0.elapsed_time(ev1)
    edges = [ev0] + marks + [ev1]
    counts = [chunk] * len(marks) + [args.steps - chunk * len(marks)]
    per_step = sorted(a.elapsed_time(b) / c for a, b, c in zip(edges[:-1], edges[1:], counts) if c > 0)
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = f.status()

    # kernel durations from HIP events on the filter's stream (separate pass: events perturb the timed loop)
    k_ev = min(args.steps, 100)
    f.timing_reset()
    f.timing(True)
    for _ in range(k_ev):
        step()
    sync()
    ms, cnt = f.timing_read()
    f.timing(False)
    avg_ms = ms / np.maximum(cnt, 1)

    # N > 1: where a step's time goes ("report resample-with-migration time separately") — the phases
    # run back to back with a device synchronisation after each, so the parts add up to more than a pipelined step
    breakdown = None
    if multi:
        k_bd = min(args.steps, 50)

        def tick():
            torch.cuda.synchronize()
            return time.perf_counter()

        if gathered:
            acc = {"local_step_into_rows": 0.0, "all_gather_rows": 0.0, "normalise+indices+import": 0.0}
        else:
            acc = {"local_step": 0.0, "all_gather": 0.0, "resample_begin": 0.0, "all_to_all": 0.0, "resample_end": 0.0}
        for _ in range(k_bd):
            t_a = tick()
            if gathered:
                rows = shard.step_local_rows(control, d_noise.data_ptr(), d_z.data_ptr(), M)
                t_b = tick()
                allrows = sf._gather_rows(rows)
                t_c = tick()
                shard.resample_gathered(allrows, u, world, rank, True, False)
                t_d = tick()
                ts_ = (t_b - t_a, t_c - t_b, t_d - t_c)
            else:
                shard.step_local_dev(control, d_noise.data_ptr(), d_z.data_ptr(), M)
                t_b = tick()
                allw = sf.gather_logweights()
                t_c = tick()
                sc, rc, send, _ = shard.resample_begin(u, world, rank, allw)
                t_d = tick()
                recv = sf._exchange(send[:sum(sc)], sc, rc, shard.pack_bytes())
                t_e = tick()
                shard.resample_end(recv)
                t_f = tick()
                ts_ = (t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, t_f - t_e)
            for key, dt_ in zip(acc, ts_):
                acc[key] += dt_
        breakdown = {key: 1e6 * v / k_bd for key, v in acc.items()}
        sync()

    # the same loop with the reference's trigger instead of a forced resample (nEff <= resample_threshold)
    unforced = None
    if not multi:
        k_un = min(args.steps, 200)
        for _ in range(5):
            f.step_dev(control, d_noise.data_ptr(), d_z.data_ptr(), M, u, force_resample=False)
        sync()
        tu = time.perf_counter()
        for _ in range(k_un):
            f.step_dev(control, d_noise.data_ptr(), d_z.data_ptr(), M, u, force_resample=False)
        sync()
        unforced = k_un / (time.perf_counter() - tu)

    # stage breakdown from the diagnostic instantiation of the update kernel: in-kernel s_memrealtime
    # stamps per workgroup (shares of the per-particle critical path; the stamped kernel is not the timed one)
    stages = None
    if not multi and args.config != 5:
        names = ["classify+ekf", "normalisers", "nondetect_emit", "detect_emit", "finalise+births", "sort", "merge_rounds",
                 "sort_by_seed", "segments", "moment_matching", "append"]
        f.debug(2)
        for _ in range(2):
            f.update(w["z"][0])
        f.sync()
        st_ = f.stamps().astype(np.int64)
        dd = np.diff(st_[:, :12], axis=1) * 0.01
        stages = {nm: float(dd[:, k].mean()) for k, nm in enumerate(names)}
        stages["workgroup_total"] = float(((st_[:, 11] - st_[:, 0]) * 0.01).mean())
        f.debug(0)

    # in-run HBM ceiling: a device-to-device copy of 1 GiB on the same stream, read + write bytes
    copy_gbs = None
    if not multi:
        src_t = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst_t = torch.empty_like(src_t)
        dst_t.copy_(src_t)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(ts)
        for _ in range(5):
            dst_t.copy_(src_t)
        c1.record(ts)
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src_t.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src_t, dst_t

    if rank == 0:
        b_step = S.algorithmic_bytes(N, G, M)               # per launch of the update+merge kernel (one shard)
        b_min = N * (28 * G + 28 * G + 32)                   # compulsory traffic
        if args.config == 5:                                 # CPHD: + one cardinality row read and written per particle
            b_step += N * 2 * 4 * 256
            b_min += N * 2 * 4 * 256
        pair_ms = avg_ms[P._lib.K_UPDATE_MERGE]              # event pair around every launch (separate pass)
        one_launch_per_step = not multi and cnt[P._lib.K_WEIGHTS] == 0 and cnt[P._lib.K_PREDICT] == 0
        # when the whole step is ONE launch of the dominant kernel (fused step), the events bracketing the timed
        # region give its average duration directly (launch-to-launch), free of the pair's marker packets
        ker_ms = gpu_region_ms / args.steps if one_launch_per_step else pair_ms
        ker_s = ker_ms * 1e-3
        achieved = b_step / ker_s / 1e9 if ker_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the PMC passes (tools/pmc_traffic.sh writes the
        # summary; rocprofv3 cannot run inside the bench) — null if this configuration was not profiled
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic_cfg%d.json" % args.config)
        if not multi and os.path.exists(tf):
            try:
                traffic = json.load(open(tf))["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "PHD-update steps/sec at N_particles x N_gauss x N_meas",
            "value": world * args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_gpu_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %d particles x %d Gaussians/particle x %d meas/step per GPU, "
                                   "Ackerman motion, forced resample every step, frozen snapshot" % (args.config - 1, N, G, M),
                       "particles_total": N * world, "gaussians_per_particle": G, "measurements_per_step": M,
                       "value_counts": "shard-steps (ranks x steps) per second",
                       "max_survivors": st["max_survivors"], "max_map": st["max_map"],
                       "steps_per_s_unforced_resample": unforced, "multi_gpu_phase_us_rank0": breakdown,
                       "multi_gpu_exchange": (("gathered" if gathered else "alltoall") if multi else None),
                       **({"one_rank_rccl_dry_run": True} if one_rank else {})},
            "stages_us_per_workgroup": stages,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "phd_update_merge_kernel", "kernel_avg_us": 1e3 * ker_ms,
                         "kernel_avg_us_event_pairs": 1e3 * pair_ms, "gpu_region_ms_per_step": gpu_region_ms / args.steps,
                         "algorithmic_bytes_per_launch": b_step, "compulsory_bytes_per_launch": b_min,
                         "device_copy_ceiling_gbs": copy_gbs,
                         "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "other_kernels_avg_us": {"phd_predict_kernel": 1e3 * avg_ms[P._lib.K_PREDICT],
                                                  "phd_weights_kernel": 1e3 * avg_ms[P._lib.K_WEIGHTS]}},
        }
        if not multi and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(w, args.config, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    f.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
