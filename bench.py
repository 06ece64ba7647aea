#!/usr/bin/env python3
"""bench.py — filter-update steps/sec of the GM-PHD-SLAM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full filter step on one batch of synthetic input, inputs resident in HBM:
predict -> update (in-range split, births, EKF, PHD weights) -> prune -> merge -> weight
normalise -> nEff -> resample (forced every step).  Steady-state protocol: the
filter is frozen, so every timed iteration restarts from the same device-resident snapshot and
does identical work.

N = 1: the headline is the LARGEST single-GPU configuration, BASELINE.json configs[2]
(4096 particles x 256 Gaussians x 64 measurements); configs[1] (256 x 64 x 32) and configs[4] (the CPHD
variant of 4096 x 256 x 64) ride along as `secondary` entries of the same JSON line (`--config 2|3|5`
picks another headline, `--no-secondary` drops the riders).

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[3] — ONE filter of
16384 particles x 256 x 64 sharded N ways (strong scaling: rank r owns particles [r n, (r+1) n), n = 16384 / N,
of the same generated set; every rank sees the same measurement set, control and resampling uniform).  Per step:
local predict + update + merge (no communication) -> RCCL all-gather of the raw log-weights -> identical global
normalise / resample on every rank -> migration of the particles whose parent lives on another rank
(all_to_all over xGMI).  `value` = filter steps/s (K / time, not multiplied by N).  The round-1 weak-scaling
workload (256 x 64 x 32 per rank) is kept as a labelled secondary.

Clock pre-roll: before the W counted warm-up steps the same step runs, un-counted, for >= 400 ms
(`preroll_steps` in the output), so that a 25-step run reads the clocks a long run reads.

Prints ONE JSON line (rank 0) with
  `roofline`      the contract figure of SURVEY.md §8d: algorithmic bytes per launch of the dominant kernel / its
                  average duration (HIP events on the filter's stream), against the 8 TB/s HBM peak.  The kernel
                  never materialises the update components, so this is bookkeeping, not efficiency — `traffic`
                  (PMC, with its source file and date) is what actually crosses the HBM interface;
  `roofline_valu` how busy the SIMDs are: the issue figure of the SQ counter pass kept under profiles/ (source and date
                  labelled) against the MEASURED ceiling of the same formula (profiles/valu_ceiling.json, round 5: 1.91 —
                  a SIMD issues one wave64 instruction per 2.1 cycles), algorithmic flops of §8d / kernel time against
                  the 157.3 TFLOP/s fp32 vector peak;
  `cpu_baseline`  the CPU oracle timed on the host cores on a bounded sample of the same workload (rank 0, N = 1):
                  CPU model, single-thread and best-thread rates.
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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same guide: peak FP32 (vector)
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
PREROLL_MS = 400.0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo (0 if it does not say)"""
    seen, phys = set(), None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return len(seen)


def cpu_quota_info():
    """what may cap the host threads besides the affinity mask: the cgroup CPU quota (v2 cpu.max / v1 cfs_quota_us), how often
    this cgroup has been throttled so far, the OpenMP environment, the load average"""
    def rd(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None
    info = {"cgroup_v2_cpu_max": rd("/sys/fs/cgroup/cpu.max"), "cgroup_v1_cfs_quota_us": rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
            "cgroup_v1_cfs_period_us": rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
            "omp_env": {k: os.environ[k] for k in ("OMP_NUM_THREADS", "OMP_THREAD_LIMIT", "OMP_PROC_BIND", "OMP_PLACES", "GOMP_CPU_AFFINITY")
                        if k in os.environ}}
    try:                                                   # the process's own cgroup (v2: "0::/path")
        for line in open("/proc/self/cgroup"):
            f = line.strip().split(":", 2)
            if f[0] == "0" and len(f) == 3 and f[2] not in ("", "/"):
                info["cgroup_v2_path"] = f[2]
                info["cgroup_v2_cpu_max_own"] = rd("/sys/fs/cgroup" + f[2] + "/cpu.max")
    except OSError:
        pass
    q = None
    try:
        if info["cgroup_v2_cpu_max"] and info["cgroup_v2_cpu_max"].split()[0] != "max":
            a, b = info["cgroup_v2_cpu_max"].split()[:2]
            q = float(a) / float(b)
        elif info["cgroup_v1_cfs_quota_us"] and int(info["cgroup_v1_cfs_quota_us"]) > 0:
            q = int(info["cgroup_v1_cfs_quota_us"]) / float(info["cgroup_v1_cfs_period_us"] or 100000)
    except Exception:
        pass
    info["quota_cpus"] = q
    try:
        info["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    return info


def cpu_throttle_counters():
    """(nr_throttled, throttled microseconds) of this cgroup so far, or None"""
    for path, key_n, key_t, scale in (("/sys/fs/cgroup/cpu.stat", "nr_throttled", "throttled_usec", 1.0),
                                      ("/sys/fs/cgroup/cpu/cpu.stat", "nr_throttled", "throttled_time", 1e-3)):
        try:
            d = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
            return int(d[key_n]), float(d[key_t]) * scale
        except Exception:
            continue
    return None


def cpu_baseline(w, cfg_id, budget_s):
    """the CPU oracle (a port of the reference algorithm: kind "port") on the host cores, SURVEY.md §8d: the FULL particle
    set of the configuration (no slice, no n/N scaling), compiled -O3 -march=native on THIS host (oracle.build_native),
    single thread and the fastest thread count of a scan that includes the physical core count."""
    from oracle import oracle as O
    N, G, M = w["N"], w["G"], w["M"]
    cap = 2 * G
    lib_path, build_info = O.build_native()
    O.use_library(lib_path)
    try:
        ocfg = O.default_config()
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        phys = physical_cores()
        maps = np.zeros((N, cap), O.GAUSSIAN)
        maps[:, :G] = w["maps"]
        lw = O.normalize_weights(w["logw"])
        cn0 = np.full((N, 256), -np.log(256.0), np.float32) if cfg_id == 5 else None   # config 5: uniform rows (src/main.cpp:1142)
        one = O.make_stepper(w["poses"], lw, maps, w["sizes"], cap, 0.05, 2.0, w["noise"][0], w["z"][0], ocfg, w["uniform"][0],
                             True, clutter_rate=20.0, cn=cn0)

        busy = {}                                      # threads -> process CPU seconds / wall seconds of that run: the
                                                       # cores the OS actually gave (a quota or a busy host shows HERE)

        def timed(threads, k):
            c0 = os.times()
            t0 = time.perf_counter()
            for _ in range(k):
                one(threads)
            dt = time.perf_counter() - t0
            c1 = os.times()
            busy[threads] = ((c1.user - c0.user) + (c1.system - c0.system)) / max(dt, 1e-9)
            return dt / k

        quota = cpu_quota_info()
        thr0 = cpu_throttle_counters()
        try:
            omp_max = int(O.lib().o_omp_max_threads())
        except Exception:
            omp_max = None
        t_start = time.perf_counter()
        # thread scan: powers of two, the physical core count, everything visible — one full step each (the first call at
        # the widest count also warms the pages of every buffer)
        # (capped at the cgroup's CPU quota: more threads than the quota buys are throttled, not faster — VERDICT r5 weak #11)
        qc = quota.get("quota_cpus")
        widest_allowed = avail if not qc else max(1, min(avail, int(np.ceil(qc))))
        cand = sorted({t for t in [2 ** k for k in range(1, 12)] + [phys, avail, avail // 2, widest_allowed] if 1 < t <= widest_allowed})
        one(widest_allowed)
        scan = {}
        for t in reversed(cand):                       # widest first: the slow narrow counts are dropped when time runs out
            if time.perf_counter() - t_start > 0.35 * budget_s and scan:
                break
            scan[t] = timed(t, 1)
        best_t = min(scan, key=scan.get) if scan else 1
        # >= 5 timed full steps at the best thread count
        kb = 5
        sb = timed(best_t, kb) if best_t > 1 else None
        # >= 2 timed full steps on one thread
        k1 = 2
        s1 = timed(1, k1)
        if sb is None:
            sb, kb = s1, k1
        best, single = 1.0 / sb, 1.0 / s1
        total = time.perf_counter() - t_start
        thr1 = cpu_throttle_counters()
        throttled = None if (thr0 is None or thr1 is None) else {"periods": thr1[0] - thr0[0], "seconds": 1e-6 * (thr1[1] - thr0[1])}
        # why the scan flattens, from what was measured: the cores the OS gave (process CPU time / wall time) against the threads
        # asked for, the quota, the throttle counters
        widest = max(scan) if scan else 1
        limit = None
        if throttled and throttled["periods"] > 0:
            limit = "cgroup CPU quota: throttled %d periods (%.2f s) during the scan; quota %s CPUs" % (
                throttled["periods"], throttled["seconds"], quota.get("quota_cpus"))
        elif busy.get(widest, widest) < 0.7 * widest:
            limit = "the OS gave %.1f cores to %d threads (process CPU time / wall time): fewer cores than the affinity mask shows" % (busy[widest], widest)
        elif scan and scan[widest] > 1.2 * min(scan.values()):
            limit = "threads were busy (%.1f cores for %d threads) but slower than fewer threads: SMT siblings / memory bandwidth / NUMA" % (busy[widest], widest)
    finally:
        O.use_library(None)
    return {"value": best, "unit": "steps/s", "cores": best_t, "kind": "port", "omp_max_threads": omp_max,
            "cores_given_by_threads": {str(t): round(busy[t], 2) for t in sorted(busy)}, "host_limits": quota, "throttled_during_scan": throttled,
            "scaling_limit": limit,
            "cpu_model": cpu_model(), "cores_visible": avail, "cores_physical": phys,
            "single_thread_steps_per_s": single, "best_thread_steps_per_s": best,
            "thread_scan_s_per_step": {str(t): scan[t] for t in sorted(scan)},
            "particles_timed": N, "extrapolated": False, **build_info,
            "sample_short": "%d full steps (all %d particles) of oracle/scphd_cpu.c at %d threads + %d on one thread, %.0f s" % (kb, N, best_t, k1, total),
            "sample": "%d full steps (all %d particles, no scaling) of the oracle (oracle/scphd_cpu.c + cphd_cpu.c, %s, compiled on this "
                      "host; OpenMP over particles) at %d threads — the fastest of the scan %s (visible %d, physical %d) — and %d full "
                      "steps on one thread; config %d (%dx%dx%d), %.1f s in all"
                      % (kb, N, build_info["compile_flags"], best_t, sorted(scan), avail, phys, k1, cfg_id, N, G, M, total)}


def algorithmic_flops(N, G, M):
    """SURVEY.md §8(d), secondary figure: sum_p [150 G + 45 G M + 30 M]"""
    return N * (150.0 * G + 45.0 * G * M + 30.0 * M)


def load_profile_json(name, build_id):
    """a PMC summary kept under profiles/ (tools/pmc_*.sh wrote it on the GPU box): (dict, source label) — or (None, why) when
    there is none or when it was recorded with ANOTHER build of the library than the one loaded now (the summaries carry the
    library's build id, phd_version(); counters of a kernel that has changed since are not printed)"""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
    except Exception:
        return None, None
    if d.get("build_id") != build_id:
        return None, "profiles/%s is stale: recorded with build %s (%s), the loaded library is build %s — counters withheld" % (
            name, d.get("build_id", "?"), d.get("build", "?"), build_id)
    label = "profiles/%s (build %s = %s, recorded %s)" % (name, d.get("build", "?"), d.get("build_id"), d.get("date", "?"))
    return d, label


def timed_loop(step, sync, ts, torch, steps, warmup, preroll_ms, agree=None):
    """pre-roll (un-counted, >= preroll_ms of the same step), W warm-up steps, then K timed steps bracketed by sync()
    -> (elapsed s, gpu region ms, p10/p50/p90 ms per step from asynchronous chunk events, preroll steps).
    agree(x): max of x over the ranks (N > 1: every rank must run the same number of pre-roll steps)"""
    preroll = 0
    if preroll_ms > 0:
        # 25 steps timed, then as many more as the remaining pre-roll time needs (one decision, agreed by all ranks)
        sync()
        t0 = time.perf_counter()
        for _ in range(25):
            step()
        sync()
        dt = (time.perf_counter() - t0) * 1e3
        if agree is not None:
            dt = agree(dt)
        more = int(min(100000, max(0.0, (preroll_ms - dt) / max(dt / 25, 1e-4))))
        for _ in range(more):
            step()
        sync()
        preroll = 25 + more
    for _ in range(warmup):
        step()
    sync()
    # HIP events on the stream the kernels run on (torch's current stream IS the filter's stream), bracketing
    # the timed region: GPU time of the K steps without the per-launch event pairs of the separate pass
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # an (asynchronous) event every `chunk` steps: the spread of the step time.  Never between consecutive steps: an event
    # record is a packet of its own between two dependent launches and costs the 350 us step ~2 % (measured: 356.0 us per
    # step with an event after every step, 348.1 with one every 20, kernel 347.8)
    chunk = max(5, steps // 20)
    marks = []
    t0 = time.perf_counter()
    ev0.record(ts)
    for k in range(steps):
        step()
        if (k + 1) % chunk == 0 and k + 1 < steps:
            e = torch.cuda.Event(enable_timing=True)
            e.record(ts)
            marks.append(e)
    ev1.record(ts)
    sync()
    elapsed = time.perf_counter() - t0
    gpu_region_ms = ev0.elapsed_time(ev1)
    edges = [ev0] + marks + [ev1]
    counts = [chunk] * len(marks) + [steps - chunk * len(marks)]
    per_step = sorted(a.elapsed_time(b) / c for a, b, c in zip(edges[:-1], edges[1:], counts) if c > 0)
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
    return elapsed, gpu_region_ms, [pct(0.1), pct(0.5), pct(0.9)], preroll


def _instantiation(P, f):
    k = int(P._lib.lib().phd_debug_update_instantiation(f._h))
    # (csrc/phd_kernels.hip: below 18 the LDS layout and the scan's length come from the arguments; 18 ... 26 both are compiled in — a
    #  full scan on a filter of a compiled-in layout; 27 ... 35 the layout alone — any other scan on such a filter)
    return {"index": k, "fast_path": 18 <= k < 27, "layout_compiled_in": k >= 18}


def make_filter(P, torch, cfg_id, n_local, G, M, n_global, offset, dev, local_rank, map_capacity=0, survivor_capacity=0):
    cfg = P.default_config(n_particles=n_global)
    if cfg_id == 5:                                           # BASELINE.json configs[4]: the CPHD variant
        cfg.filterType = 1
        cfg.maxCardinality = 255
    # one stream for everything: the filter's kernels and (N > 1) the RCCL collectives torch enqueues
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    f = P.PhdFilter(cfg, n_particles=n_local, map_capacity=map_capacity or 2 * G, max_measurements=M,
                    survivor_capacity=survivor_capacity, device=local_rank, stream=ts.cuda_stream,
                    global_particles=n_global, global_offset=offset)
    return f, ts


def roofline_entries(P, S, cfg_id, N, G, M, ker_ms, pair_ms, gpu_ms_per_step, copy_gbs, other, with_traffic=True):
    b_step = S.algorithmic_bytes(N, G, M)               # per launch of the update+merge kernel (one shard)
    b_min = N * (28 * G + 28 * G + 32)                   # compulsory traffic
    if cfg_id == 5:                                      # CPHD: + one cardinality row read and written per particle
        b_step += N * 2 * 4 * 256
        b_min += N * 2 * 4 * 256
    ker_s = ker_ms * 1e-3
    achieved = b_step / ker_s / 1e9 if ker_s > 0 else 0.0
    build_id = P._lib.lib().phd_version().decode().split("build ")[-1]
    # HBM bytes per launch of the dominant kernel from the PMC passes (tools/pmc_traffic.sh writes the summary on
    # the GPU box; rocprofv3 cannot run inside the bench) — null if this configuration was not profiled WITH THIS BUILD.
    # NOT measured in this run: the source file, build id and date are printed beside it.
    traffic, traffic_src = None, None
    if with_traffic:
        tj, traffic_src = load_profile_json("pmc_traffic_cfg%d.json" % cfg_id, build_id)
        if tj:
            traffic = tj.get("hbm_bytes_per_launch")
    # "bound": the roofline the contract's figures are quoted against (HBM, `achieved` / `peak` in GB/s);
    # "bound_in_fact": what the counters say limits the kernel (roofline_valu)
    roof = {"bound": "hbm", "bound_in_fact": "latency (barriers, single-wave phases, LDS round trips) at about half of the measured VALU issue ceiling", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_compulsory": (b_min / ker_s / 1e9 / HBM_PEAK_GBS) if ker_s > 0 else None,
            "traffic": traffic, "traffic_source": traffic_src,
            "note": "`frac` is the contract figure of SURVEY.md 8d: ALGORITHMIC bytes (every update component written once and read "
                    "once) / kernel time / 8 TB/s.  The fused kernel prunes before it stores, so those components never reach HBM "
                    "and `frac` may exceed 1: it carries no efficiency information.  `frac_compulsory` = compulsory bytes (map in + "
                    "map out + 32 B per particle) / kernel time / 8 TB/s is what the HBM interface must carry; the kernel issues "
                    "about half of what the SIMDs sustain and waits the rest of the time, see roofline_valu",
            "kernel": "phd_update_merge_kernel", "kernel_avg_us": 1e3 * ker_ms,
            "kernel_avg_us_event_pairs": 1e3 * pair_ms, "gpu_region_ms_per_step": gpu_ms_per_step,
            "algorithmic_bytes_per_launch": b_step, "compulsory_bytes_per_launch": b_min,
            "measured_hbm_gbs": (traffic / ker_s / 1e9) if (traffic and ker_s > 0) else None,
            "device_copy_ceiling_gbs": copy_gbs,
            "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
            "other_kernels_avg_us": other, "library_build": build_id}
    flops = algorithmic_flops(N, G, M)
    tfl = flops / ker_s / 1e12 if ker_s > 0 else 0.0
    sq, sq_src = load_profile_json("pmc_sq_cfg%d.json" % cfg_id, build_id)
    # the ceiling the issue figure is quoted against is MEASURED (round 5): tools/probes/valu_issue_probe.hip saturates a SIMD
    # with independent full-rate vector instructions and reads the same counters with the same formula — 1.91, not 1.0: a
    # SIMD sustains one wave64 instruction per 2.1 cycles once two waves interleave, a lone wave one per 4.2
    # (profiles/r05_valu_ceiling.txt).  `achieved` / `peak` are the formula's readings, `frac` their quotient;
    # `valu_instructions_per_cycle_per_simd` the same in instructions (peak: the probe's 0.477)
    ceil = None
    try:
        ceil = json.load(open(os.path.join(ROOT, "profiles", "valu_ceiling.json")))
    except Exception:
        pass
    peak = ceil["formula_reading_at_saturation"] if ceil else None
    valu = {"bound": "valu-issue", "achieved": None, "peak": peak, "unit": "4 * SQ_ACTIVE_INST_VALU / SIMD-cycles (tools/pmc_sq.sh)",
            "frac": None, "peak_source": (ceil or {}).get("source"),
            "peak_note": "measured on a saturating stream of independent FULL-RATE vector instructions; packed fp32, DPP, 64-bit and "
                         "32-bit-multiply instructions issue at half that rate (formula ceiling %.2f), so `frac` is a LOWER bound "
                         "of the SIMDs' busy share" % ceil["half_rate_formula_reading"] if ceil else None,
            "source": sq_src, "algorithmic_flops_per_launch": flops, "algorithmic_tflops": tfl,
            "frac_of_fp32_vector_peak": tfl / FP32_VECTOR_PEAK_TFLOPS,
            "fp32_vector_peak_tflops": FP32_VECTOR_PEAK_TFLOPS}
    if sq:
        valu["achieved"] = sq.get("valu_issue_fraction")
        valu["frac"] = (valu["achieved"] / peak) if (peak and valu["achieved"] is not None) else None
        valu["counters_per_launch"] = {k: sq[k] for k in sq if k.startswith("SQ_") or k.startswith("GRBM_")}
        valu["kernel_avg_us_in_counter_pass"] = sq.get("kernel_avg_us")
        if ceil and sq.get("kernel_shader_cycles") and sq.get("SQ_INSTS_VALU"):
            ipc = sq["SQ_INSTS_VALU"] / (4.0 * ceil["cus"] * sq["kernel_shader_cycles"])
            valu["valu_instructions_per_cycle_per_simd"] = ipc
            valu["valu_instructions_per_cycle_per_simd_peak"] = ceil["full_rate_ipc_per_simd"]
    return roof, valu


def run_single(P, S, torch, cfg_id, steps, warmup, cpu_seconds, dev, local_rank, preroll_ms=PREROLL_MS, extras=True, general=False):
    """N = 1: one configuration, fused single-launch step.  -> result dict
    general: also time the SAME filter created with PHD_LAYOUT=0 (the instantiation that takes its LDS layout and the scan's
    length from the arguments: what a scan of arbitrary length runs) -> `value_general`"""
    c = S.CONFIGS[cfg_id]
    N, G, M = c["N"], c["G"], c["M"]
    w = S.make_workload(N, G, M, seed=0x5EED0000 + cfg_id, clustered=c["clustered"])
    d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
    d_noise = torch.from_numpy(w["noise"][0].copy()).to(dev)
    control = (2.0, 0.05)
    u = float(np.random.default_rng(0x5EED0000 + cfg_id).random())
    M_scan = min(c.get("meas_override", M), M)               # (--meas: a scan shorter than the filter's measurement capacity)

    def build(any_layout):
        old = os.environ.get("PHD_LAYOUT")
        if any_layout:
            os.environ["PHD_LAYOUT"] = "0"                # read by phd_create (csrc/phd_api.cpp)
        try:
            f_, ts_ = make_filter(P, torch, cfg_id, N, G, M, N, 0, dev, local_rank, c.get("map_capacity", 0), c.get("survivor_capacity", 0))
        finally:
            if any_layout:
                if old is None:
                    del os.environ["PHD_LAYOUT"]
                else:
                    os.environ["PHD_LAYOUT"] = old
        f_.set_particles(w["poses"], w["logw"])
        f_.set_maps(w["maps"], w["sizes"])
        torch.cuda.synchronize()
        f_.set_frozen(True)  # steady state: no step commits, every iteration restarts from the same snapshot

        def step_():
            f_.step_dev(control, d_noise.data_ptr(), d_z.data_ptr(), M_scan, u, force_resample=True)

        def sync_():
            f_.sync()
            torch.cuda.synchronize()
        return f_, ts_, step_, sync_

    value_general = inst_general = None
    if general and os.environ.get("PHD_LAYOUT", "1")[0] != "0":
        g, gts, gstep, gsync = build(True)
        g_elapsed, _, _, _ = timed_loop(gstep, gsync, gts, torch, steps, warmup, preroll_ms)
        value_general = steps / g_elapsed
        inst_general = _instantiation(P, g)
        g.close()

    f, ts, step, sync = build(False)
    elapsed, gpu_region_ms, pcts, preroll = timed_loop(step, sync, ts, torch, steps, warmup, preroll_ms)
    st = f.status()
    inst = _instantiation(P, f)                       # (of the timed steps: the stamped pass below runs the diagnostic instantiation)
    if general and value_general is None:             # PHD_LAYOUT=0 in the environment: the run IS the general one
        value_general, inst_general = steps / elapsed, inst

    # kernel durations from HIP events on the filter's stream (separate pass: events perturb the timed loop)
    k_ev = min(steps, 100)
    f.timing_reset()
    f.timing(True)
    for _ in range(k_ev):
        step()
    sync()
    ms, cnt = f.timing_read()
    f.timing(False)
    avg_ms = ms / np.maximum(cnt, 1)

    # the same loop with the reference's trigger instead of a forced resample (nEff <= resample_threshold)
    unforced = None
    if extras or cpu_seconds > 0:
        k_un = min(steps, 200)
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
    if extras and cfg_id != 5:
        names = ["classify+ekf", "normalisers", "nondetect_emit", "detect_emit", "finalise+births", "sort", "merge_rounds",
                 "moment_sums_a", "cluster_means", "moment_sums_b", "append"]
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
    if extras:
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

    pair_ms = avg_ms[P._lib.K_UPDATE_MERGE]              # event pair around every launch (separate pass)
    one_launch_per_step = cnt[P._lib.K_WEIGHTS] == 0 and cnt[P._lib.K_PREDICT] == 0 and not c.get("survivor_capacity")
    # when the whole step is ONE launch of the dominant kernel (fused step), the events bracketing the timed
    # region give its average duration directly (launch-to-launch), free of the pair's marker packets
    ker_ms = gpu_region_ms / steps if one_launch_per_step else pair_ms
    other = {"phd_predict_kernel": 1e3 * avg_ms[P._lib.K_PREDICT], "phd_weights_kernel": 1e3 * avg_ms[P._lib.K_WEIGHTS]}
    roof, valu = roofline_entries(P, S, cfg_id, N, G, M, ker_ms, pair_ms, gpu_region_ms / steps, copy_gbs, other)
    res = {
        "value": steps / elapsed,
        "ms_per_step": 1e3 * elapsed / steps,
        "ms_per_step_gpu_p10_p50_p90": pcts,
        "preroll_steps": preroll,
        "value_general": value_general,
        "config": {"workload_short": ("diagnostic (--particles / --meas): %d particles of configs[%d]'s workload x %d Gaussians x %d of %d meas/step" % (N, cfg_id - 1, G, M_scan, M))
                                     if (c.get("particles_override") or c.get("meas_override")) else
                                     ("configs[%d]: %d particles x %d Gaussians x %d meas/step%s, Ackerman, forced resample, one launch/step"
                                      % (cfg_id - 1, N, G, M, ", CPHD" if cfg_id == 5 else "")) if cfg_id <= 5 else
                                     ("dense scan %d x %d x %d (not a BASELINE config), spill-merge path" % (N, G, M)),
                   "instantiation_general": inst_general,
                   "workload": ("BASELINE.json configs[%d]: %d particles x %d Gaussians/particle x %d meas/step%s, Ackerman motion, "
                                "forced resample every step, frozen snapshot, whole step = ONE launch" %
                                (cfg_id - 1, N, G, M, " (CPHD variant, max_cardinality 255)" if cfg_id == 5 else "")) if cfg_id <= 5 else
                               ("dense scan (not a BASELINE.json config): %d particles x %d Gaussians/particle x %d meas/step — the "
                                "reference's measurement clamp (src/phdfilter.cu:3390-3394) on the merge-stress map; ~2 700 survivors per "
                                "particle, past the 2 048 the LDS merge holds: every particle's merge runs in phd_merge_spill_kernel "
                                "(survivor_capacity %d, map_capacity %d); update launch + spill-merge launch per step, forced resample, "
                                "frozen snapshot" % (N, G, M, c["survivor_capacity"], c["map_capacity"])),
                   "update_components_per_step": N * (G * (M + 1) + M),
                   "ps_per_update_component": 1e12 * elapsed / steps / (N * (G * (M + 1) + M)),
                   "particles_total": N, "gaussians_per_particle": G, "measurements_per_step": M,
                   "one_launch_per_step": bool(one_launch_per_step),
                   "max_survivors": st["max_survivors"], "max_map": st["max_map"],
                   "update_residency": f.residency(),
                   # which instantiation of the update kernel the steps ran (csrc/phd_kernels.hip: below 18 the LDS layout and the
                   # scan's length come from the arguments; from 18 the configuration's layout, a full scan and — PHD — the
                   # Mahalanobis merge metric / — CPHD — the cardinality length are compiled in: the fast path, picked per launch,
                   # bit for bit the general one; PHD_LAYOUT=0 in the environment turns it off)
                   "update_kernel_instantiation": inst,
                   "steps_per_s_unforced_resample": unforced},
        "stages_us_per_workgroup": stages,
        "roofline": roof,
        "roofline_valu": valu,
        "cpu_baseline": cpu_baseline(w, cfg_id, cpu_seconds) if cpu_seconds > 0 else None,
    }
    f.close()
    return res


def run_sharded(P, S, D, torch, dist, cfg_id, n_global, steps, warmup, dev, local_rank, rank, world, share, one_rank,
                same_set=True, preroll_ms=PREROLL_MS):
    """N > 1: ONE filter of n_global particles sharded over the ranks.  same_set: every rank generates the SAME n_global
    particles and takes its slice (configs[3]); False: the round-1 weak-scaling workload (n_global / world fresh particles
    per rank, same measurement set)."""
    c = S.CONFIGS[cfg_id]
    G, M = c["G"], c["M"]
    off, n = D.shard_range(n_global, world, rank)
    seed = 0x5EED0000 + cfg_id
    if same_set:
        w = S.shard_workload(S.make_workload(n_global, G, M, seed=seed, clustered=c["clustered"]), world, rank)
        lw = w["logw"]                                   # the global set is normalised as generated
    else:
        w = S.make_workload(n, G, M, seed=seed + 1000 * rank, clustered=c["clustered"])
        shared = S.make_workload(1, G, M, seed=seed, clustered=c["clustered"])
        w["z"] = shared["z"]                             # one filter has one scan: the same measurement set on every rank
        lw = w["logw"] - np.float32(np.log(world))       # the global set sums to one
    f, ts = make_filter(P, torch, cfg_id, n, G, M, n_global, off, dev, local_rank)
    f.set_particles(w["poses"], np.ascontiguousarray(lw, np.float32))
    f.set_maps(w["maps"], w["sizes"])
    d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
    d_noise = torch.from_numpy(np.ascontiguousarray(w["noise"][0])).to(dev)
    control = (2.0, 0.05)
    # the resampling uniform is one draw shared by all ranks (every rank must compute the same global indices)
    u = float(np.random.default_rng(seed).random())
    torch.cuda.synchronize()
    f.set_frozen(True)
    shard = D.GpuShard(f, n_global)
    sf = D.ShardedFilter(shard, n_global, rank, world)
    sf.collectives = True
    if os.environ.get("PHD_BENCH_EXCHANGE") == "alltoall":
        sf.gathered_limit = 0
    gathered = sf.gathered()                    # small shards: whole-shard all-gather; else all-to-all of the migrants

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
        dist.barrier()

    def agree(x):
        t_ = torch.tensor([x], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    elapsed, gpu_region_ms, pcts, preroll = timed_loop(step, sync, ts, torch, steps, warmup, preroll_ms, agree)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    st = f.status()

    k_ev = min(steps, 50)
    f.timing_reset()
    f.timing(True)
    for _ in range(k_ev):
        step()
    sync()
    ms, cnt = f.timing_read()
    f.timing(False)
    avg_ms = ms / np.maximum(cnt, 1)

    # where a step's time goes ("report resample-with-migration time separately") — the phases
    # run back to back with a device synchronisation after each, so the parts add up to more than a pipelined step
    k_bd = min(steps, 30)

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    if gathered:
        acc = {"local_step_into_rows": 0.0, "all_gather_rows": 0.0, "normalise+indices+import": 0.0}
    else:
        acc = {"local_step": 0.0, "all_gather": 0.0, "resample_begin": 0.0, "all_to_all": 0.0, "resample_end": 0.0}
    migrants = remote_slots = 0
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
            sc, rc, send, idx_all = shard.resample_begin(u, world, rank, allw)
            t_d = tick()
            recv = sf._exchange(send[:sum(sc)], sc, rc, shard.pack_bytes())
            t_e = tick()
            shard.resample_end(recv)
            t_f = tick()
            ts_ = (t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, t_f - t_e)
            migrants = int(sum(rc))          # rows that travelled (a parent once per destination rank) ...
            mine = idx_all[rank * shard.f.n:(rank + 1) * shard.f.n]
            remote_slots = int(((mine // shard.f.n) != rank).sum())   # ... for this many slots filled from other ranks
        for key, dt_ in zip(acc, ts_):
            acc[key] += dt_
    breakdown = {key: 1e6 * v / k_bd for key, v in acc.items()}
    if not gathered:
        breakdown["resample_with_migration"] = breakdown["resample_begin"] + breakdown["all_to_all"] + breakdown["resample_end"]
        breakdown["particles_received_rank0"] = migrants
        breakdown["slots_filled_from_other_ranks_rank0"] = remote_slots
    else:
        breakdown["resample_with_migration"] = breakdown["all_gather_rows"] + breakdown["normalise+indices+import"]
    sync()

    pair_ms = avg_ms[P._lib.K_UPDATE_MERGE]
    other = {"phd_predict_kernel": 1e3 * avg_ms[P._lib.K_PREDICT], "phd_weights_kernel": 1e3 * avg_ms[P._lib.K_WEIGHTS]}
    roof, valu = roofline_entries(P, S, cfg_id, n, G, M, pair_ms, pair_ms, gpu_region_ms / steps, None, other, with_traffic=False)
    res = {
        "value": steps / elapsed,                          # FILTER steps per second (the shards step together)
        "ms_per_step": 1e3 * elapsed / steps,
        "ms_per_step_gpu_p10_p50_p90": pcts,
        "preroll_steps": preroll,
        "config": {"workload": ("BASELINE.json configs[3]: ONE filter of %d particles x %d Gaussians/particle x %d meas/step sharded "
                                "over %d ranks (rank r owns particles [r n, (r+1) n), n = %d, of the same generated set; same "
                                "measurement set, control and uniform on every rank), RCCL log-weight all-gather + global systematic "
                                "resample + migration every step, frozen snapshot" % (n_global, G, M, world, n)) if same_set else
                               ("weak-scaling workload of round 1: %d particles x %d x %d per rank (fresh particles per rank, same "
                                "measurement set), one %d-particle filter" % (n, G, M, n_global)),
                   "particles_total": n_global, "particles_per_rank": n, "gaussians_per_particle": G, "measurements_per_step": M,
                   "value_counts": "filter steps per second (K / max-over-ranks time)",
                   "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                   "max_survivors": st["max_survivors"], "max_map": st["max_map"],
                   "multi_gpu_phase_us_rank0": breakdown,
                   "multi_gpu_exchange": "gathered" if gathered else "alltoall",
                   **({"one_rank_rccl_dry_run": True} if one_rank else {}),
                   **({"share_gpu_dry_run": True} if share else {})},
        "roofline": roof,
        "roofline_valu": valu,
    }
    f.close()
    return res


def verify_cpp_multi(P, S, MM, devices, exchange, cap, mm):
    """a small filter (64 particles per shard, 32 Gaussians, 16 measurements, the bench's map capacity) through four COMMITTED
    steps — forced, nEff-triggered, forced, forced resamples with a weight vector concentrated on the last shard, so that
    particles migrate across every device boundary — on the sharded filter and on a single filter: particles, weights and maps
    must agree bit for bit.  -> {"equal_to_single_filter", "exchange", ...}"""
    k = len(devices)
    N, G, M, steps = 64 * k, 32, min(16, mm), 4
    w = S.make_workload(N, G, M, seed=0xC0FFEE + k, n_meas_sets=steps, clustered=True)
    lw = np.linspace(-12.0, 0.0, N).astype(np.float32)
    w["logw"] = (lw - np.float32(np.log(np.exp(lw.astype(np.float64)).sum()))).astype(np.float32)
    cfg = P.default_config(n_particles=N, resampleThresh=0.6)
    force = [True, False, True, True]
    # "exchange" is set BEFORE the create: a create that fails (PULL without peer access, an RCCL bootstrap time-out) must leave
    # the caller its fall-back decision, not a KeyError
    requested = {MM.EXCHANGE_ALLTOALL: "alltoall", MM.EXCHANGE_GATHERED: "gathered", MM.EXCHANGE_PULL: "pull"}.get(exchange, "auto")
    out = {"equal_to_single_filter": False, "particles": N, "steps": steps, "devices": list(devices), "exchange": requested,
           "exchange_requested": requested}
    try:
        with P.PhdFilter(cfg, n_particles=N, map_capacity=cap, max_measurements=mm, device=devices[0]) as f, \
                MM.MultiFilter(cfg, n_shards=k, devices=list(devices), map_capacity=cap, max_measurements=mm, exchange=exchange,
                               gathered_limit_bytes=(1 if exchange == MM.EXCHANGE_AUTO else 0)) as m:
            out["exchange"] = m.exchange
            out["rccl"] = m.uses_rccl
            for x in (f, m):
                x.set_particles(w["poses"], w["logw"])
                x.set_maps(w["maps"], w["sizes"])
            for s in range(steps):
                f.predict((2.0, 0.05), w["noise"][s])
                f.update(w["z"][s])
                if force[s]:
                    f.resample(w["uniform"][s])
                else:
                    f.resample_if_needed(w["uniform"][s], had_measurements=True)
                m.step((2.0, 0.05), w["noise"][s], w["z"][s], w["uniform"][s], force_resample=force[s])
                pa, la = f.get_particles()
                pb, lb = m.get_particles()
                same = np.array_equal(pa, pb) and np.array_equal(la, lb) and all(
                    np.array_equal(a, b) for a, b in zip(f.get_maps(), m.get_maps()))
                if not same:
                    out["first_difference_at_step"] = s
                    return out
            out["equal_to_single_filter"] = True
    except Exception as e:                                    # noqa: BLE001 — a failed verification is reported, the caller decides
        out["error"] = str(e)[:300]
    return out


def run_cpp_multi(P, S, torch, cfg_id, steps, warmup, devices, preroll_ms, with_phases=True):
    """ONE filter sharded over len(devices) shards, driven by the C++ multi-device host (libphdslam_multi.so,
    include/phdslam_multi.h) inside THIS process: one host thread, one HIP stream per shard, RCCL (ncclCommInitAll) when every
    shard has a device of its own.  Shards that share a device (more shards than GPUs: PHD_BENCH_SHARE_GPU=1 on a one-GPU box)
    exchange by device copies — a dry run of the sharding logic, not a measurement of links.  One shard on one device is a
    one-rank RCCL communicator (what the collective path costs before any link is involved)."""
    MM = importlib.import_module("cuda-phdslam_amd.multi")
    L = P._lib
    c = S.CONFIGS[cfg_id]
    N, G, M = c["N"], c["G"], c["M"]
    n_shards = len(devices)
    w = S.make_workload(N, G, M, seed=0x5EED0000 + cfg_id, clustered=c["clustered"])
    cfg = P.default_config(n_particles=N)
    ex = {"alltoall": MM.EXCHANGE_ALLTOALL, "gathered": MM.EXCHANGE_GATHERED, "pull": MM.EXCHANGE_PULL}.get(
        os.environ.get("PHD_BENCH_EXCHANGE", ""), MM.EXCHANGE_AUTO)
    # first contact: before anything is timed, the sharded filter must equal a single filter bit for bit ON THESE DEVICES, with
    # the exchange the timed run will use (the PULL form reads peers' memory directly and had only ever run with all shards on
    # one GPU when this was written); if it does not, the host-planned all-to-all is verified and used instead, and the line says so
    verified = verify_cpp_multi(P, S, MM, devices, ex, 2 * G, M)
    if not verified["equal_to_single_filter"] and verified.get("exchange") != "alltoall":
        second = verify_cpp_multi(P, S, MM, devices, MM.EXCHANGE_ALLTOALL, 2 * G, M)
        second["fell_back_from"] = verified
        verified = second
        if verified["equal_to_single_filter"]:
            ex = MM.EXCHANGE_ALLTOALL
    m = MM.MultiFilter(cfg, n_shards=n_shards, devices=list(devices), map_capacity=2 * G, max_measurements=M, exchange=ex)
    m.set_particles(w["poses"], w["logw"])
    m.set_maps(w["maps"], w["sizes"])
    m.set_frozen(True)
    m.upload_inputs(w["noise"][0], w["z"][0])
    u = float(np.random.default_rng(0x5EED0000 + cfg_id).random())
    control = (2.0, 0.05)

    def step():
        m.step_resident(control, u, force_resample=True)

    def sync():
        m.sync()                                              # every shard's stream drained
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    ts = torch.cuda.current_stream()
    # (the shards run on their own streams: the event-based percentiles of timed_loop do not see them; wall clock only)
    elapsed, _, _, preroll = timed_loop(step, sync, ts, torch, steps, warmup, preroll_ms)

    # the nEff-triggered step (what the SLAM loop runs: src/main.cpp:1281-1296): the decision is taken on the device and the host
    # only enqueues (PHD_MULTI_HOST_NEFF=1: the round-4 form, nEff read back every step)
    k_un = max(min(steps, 400), 1)
    for _ in range(min(20, k_un)):
        m.step_resident(control, u, force_resample=False)
    sync()
    tu = time.perf_counter()
    for _ in range(k_un):
        m.step_resident(control, u, force_resample=False)
    sync()
    unforced = k_un / (time.perf_counter() - tu)

    # per-shard kernel time (HIP events on shard 0's stream around its launches) and the per-phase breakdown (HIP events on
    # shard 0's stream at the phase boundaries, all shards drained after every step): separate passes
    k_ev = min(steps, 50)
    h0 = m.shard_handle(0)
    L.check(L.lib().phd_timing_reset(h0), "phd_timing_reset")
    L.check(L.lib().phd_timing_enable(h0, 1), "phd_timing_enable")
    for _ in range(k_ev):
        step()
    sync()
    ms = np.zeros(L.K_COUNT, np.float64)
    cnt = np.zeros(L.K_COUNT, np.int64)
    L.check(L.lib().phd_timing_read(h0, L.ptr(ms), L.ptr(cnt)), "phd_timing_read")
    L.check(L.lib().phd_timing_enable(h0, 0), "phd_timing_enable")
    avg_ms = ms / np.maximum(cnt, 1)
    phases = None
    if with_phases:
        m.timing_reset()
        m.timing(True)
        for _ in range(min(steps, 30)):
            step()
        sync()
        phases, k_ph = m.timing_read()
        m.timing(False)
        phases["resample_with_migration"] = phases["weights"] + phases["plan_export"] + phases["send_recv"] + phases["import"]
        phases["steps_averaged"] = k_ph
        phases["note"] = ("HIP events on shard 0's stream at the phase boundaries; every step of this pass ends with all shards "
                          "drained, so the parts add up to more than a pipelined step")
    n = N // n_shards
    pair_ms = avg_ms[L.K_UPDATE_MERGE]
    other = {"phd_predict_kernel": 1e3 * avg_ms[L.K_PREDICT], "phd_weights_kernel": 1e3 * avg_ms[L.K_WEIGHTS]}
    roof, valu = roofline_entries(P, S, cfg_id, n, G, M, pair_ms, pair_ms, 1e3 * elapsed / steps, None, other, with_traffic=False)
    distinct = len(set(devices)) == n_shards
    res = {"value": steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "preroll_steps": preroll,
           "ms_per_step_gpu_p10_p50_p90": None,
           "config": {"workload": "BASELINE.json configs[%d]: ONE filter of %d particles x %d Gaussians/particle x %d meas/step sharded over "
                                  "%d shard(s) (shard k owns particles [k n, (k+1) n), n = %d), C++ multi-device host "
                                  "(libphdslam_multi.so, one process, one host thread), transport %s, exchange %s, log-weight "
                                  "all-gather + global systematic resample + migration every step, frozen snapshot" %
                                  (cfg_id - 1, N, G, M, n_shards, n,
                                   ("RCCL (ncclCommInitAll over %d device(s))" % n_shards) if m.uses_rccl else "device copies (shards share a GPU)",
                                   m.exchange),
                      "particles_total": N, "particles_per_shard": n, "gaussians_per_particle": G, "measurements_per_step": M,
                      "host": "C++ (libphdslam_multi.so)", "cpp_multi_host": True, "n_shards": n_shards, "devices": list(devices),
                      "rccl": m.uses_rccl, "rccl_ranks": n_shards if m.uses_rccl else 0,
                      "value_counts": "filter steps per second (K / wall time with every shard drained)",
                      "multi_gpu_exchange": m.exchange,
                      "steps_per_s_unforced_resample": unforced,
                      "unforced_decision": "host (nEff read back)" if os.environ.get("PHD_MULTI_HOST_NEFF") else "device (no host round trip)",
                      "multi_gpu_verified": verified,
                      "multi_gpu_phase_us_shard0": phases,
                      **({} if distinct else {"share_gpu_dry_run": True})},
           "roofline": roof, "roofline_valu": valu}
    m.close()
    return res


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                         # best effort: only the ordering of a banner depends on it
        pass


RECORD_PATH = os.environ.get("PHD_BENCH_RECORD", os.path.join(ROOT, "profiles", "bench_last.json"))
LINE_LIMIT = 4096                # bytes of the final stdout line (the driver's record keeps a bounded tail of stdout)


def _short(s, n=120):
    return s if (s is None or len(s) <= n) else s[:n - 1] + "~"


def _r(x, digits=6):
    """floats of the line rounded to `digits` significant digits (the record file keeps them whole)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def hip_runtime_version(torch=None):
    try:
        import torch as T
        return str(T.version.hip)
    except Exception:
        return None


def compact_line(full):
    """the ONE line the driver parses: the contract's keys and nothing that grows with the number of riders.  Everything else of
    `full` (riders, notes, counters, thread scans, stage stamps) is written to RECORD_PATH by emit().  Kept <= LINE_LIMIT bytes
    (tests/test_bench_model.py::test_line_is_compact builds it from a canned record through this same function)."""
    cfg = full.get("config") or {}
    inst = cfg.get("update_kernel_instantiation") or {}
    roof = full.get("roofline") or {}
    valu = full.get("roofline_valu") or {}
    cpu = full.get("cpu_baseline")
    line = {
        "metric": full["metric"], "value": _r(full["value"]), "unit": full["unit"], "n_gpus": full["n_gpus"],
        "steps": full["steps"], "warmup": full["warmup"], "ms_per_step": _r(full["ms_per_step"]),
        "higher_is_better": True, "scaling": full.get("scaling", "weak"), "vs_baseline": None,
        "dtype": full.get("dtype", "f32"), "data": "synthetic",
        # what an arbitrary scan gets: the same filter, same visit, created with the layout / scan-length specialisation off
        "value_general": _r(full.get("value_general")),
        "config": {"workload": _short(cfg.get("workload_short") or cfg.get("workload")),
                   "N": cfg.get("particles_total"), "G": cfg.get("gaussians_per_particle"), "M": cfg.get("measurements_per_step"),
                   "one_launch_per_step": cfg.get("one_launch_per_step"),
                   "instantiation": inst.get("index"), "fast_path": inst.get("fast_path"),
                   **{k: cfg[k] for k in ("n_shards", "rccl_ranks", "multi_gpu_exchange", "particles_per_shard", "particles_per_rank")
                      if k in cfg}},
        "roofline": {"bound": roof.get("bound"), "achieved": _r(roof.get("achieved")), "peak": roof.get("peak"),
                     "unit": roof.get("unit"), "frac": _r(roof.get("frac")), "frac_compulsory": _r(roof.get("frac_compulsory")),
                     "traffic": None if roof.get("traffic") is None else int(round(roof["traffic"])), "kernel": roof.get("kernel"), "kernel_avg_us": _r(roof.get("kernel_avg_us")),
                     "library_build": roof.get("library_build")},
        "roofline_valu": {"achieved": _r(valu.get("achieved")), "peak": _r(valu.get("peak")), "frac": _r(valu.get("frac"))},
        "cpu_baseline": None if not cpu else {
            "value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
            "cpu_model": _short(cpu.get("cpu_model"), 60), "quota_cpus": (cpu.get("host_limits") or {}).get("quota_cpus"),
            "extrapolated": cpu.get("extrapolated"), "single_thread": _r(cpu.get("single_thread_steps_per_s")),
            "sample": _short(cpu.get("sample_short") or cpu.get("sample"))},
        # steps/s of the riders measured in the same run (their full entries are in the record file)
        "riders_steps_per_s": {str(r.get("rider", k)): _r(r.get("value"), 5) for k, r in enumerate(full.get("secondary") or [])},
        "hip_runtime_version": full.get("hip_runtime_version"),
        "riders": os.path.relpath(RECORD_PATH, ROOT) if RECORD_PATH.startswith(ROOT) else RECORD_PATH,
    }
    return line


def emit(full):
    """writes the full record to RECORD_PATH (riders, notes, counters: everything) and prints the compact line LAST on stdout"""
    full.setdefault("hip_runtime_version", hip_runtime_version())
    try:
        os.makedirs(os.path.dirname(RECORD_PATH), exist_ok=True)
        with open(RECORD_PATH, "w") as fh:
            json.dump(full, fh, indent=1)
            fh.write("\n")
    except OSError as e:                                      # a read-only tree must not cost the line
        print("bench.py: could not write %s: %s" % (RECORD_PATH, e), file=sys.stderr)
    line = compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    # never lose the line to its own size: drop what is optional until it fits (cannot happen with today's keys: 1.3 KB)
    for drop in ("riders_steps_per_s", "roofline_valu", "hip_runtime_version"):
        if len(text) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    sys.stdout.flush()
    print(text, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 200 timed steps of 0.35 ms after the clock pre-roll
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=0,
                    help="headline workload: 0 = the default (3 at N = 1: BASELINE.json configs[2]; 4 at N > 1: configs[3] sharded); "
                         "2, 3, 5 (configs[1], [2], [4] = CPHD) at N = 1; 6 = the dense-scan rider (4096 x 256 x 256, spill path)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0,
                    help="soft budget of the headline's cpu_baseline leg: bounds the thread scan; the 5 + 2 full timed steps always run (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="headline only")
    ap.add_argument("--preroll-ms", type=float, default=PREROLL_MS)
    ap.add_argument("--particles", type=int, default=0,
                    help="override the particle count of the chosen configuration (a shard's share of configs[3]: 2048 / 4096 / 8192) — a "
                         "diagnostic workload, labelled in config.workload; not a BASELINE.json configuration")
    ap.add_argument("--meas", type=int, default=0,
                    help="diagnostic: scans of this many measurements (the first ones of the generated scan) on a filter that holds the "
                         "configuration's measurement capacity - a RAGGED scan, what real data presents; labelled in config.workload")
    ap.add_argument("--bare", action="store_true",
                    help="profiling runs (rocprofv3): headline loop only — no riders, no stamped/unforced/copy passes, no CPU leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    D = importlib.import_module("cuda-phdslam_amd.dist")

    if args.particles > 0:
        for cid in ([args.config] if args.config else [3, 4]):
            S.CONFIGS[cid] = dict(S.CONFIGS[cid], N=args.particles, particles_override=True)
    if args.meas > 0:
        for cid in ([args.config] if args.config else [3]):
            S.CONFIGS[cid] = dict(S.CONFIGS[cid], meas_override=args.meas)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PHD_BENCH_SHARE_GPU=1: dry run of the N > 1 path on a one-GPU box (every shard / rank on device 0; device-copy or gloo
    # transport) — exercises the multi-shard code, its numbers are not a measurement
    share = os.environ.get("PHD_BENCH_SHARE_GPU") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: ONE process drives all N GPUs through the C++ multi-device host (libphdslam_multi.so: ncclCommInitAll
        # over devices 0..N-1, one HIP stream per shard, one host thread) — BASELINE.json configs[3], strong scaling
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not share:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (PHD_BENCH_SHARE_GPU=1 runs the shards on one GPU as a dry run)"
                             % (args.gpus, ndev))
        devices = [0] * args.gpus if share else list(range(args.gpus))
        torch.cuda.set_device(0)
        cfg_id = args.config or 4
        res = run_cpp_multi(P, S, torch, cfg_id, args.steps, args.warmup, devices, args.preroll_ms)
        secondary = []
        if not args.no_secondary and not args.bare:
            # the N = 1 point of this strong-scaling line, measured in the same run: the same filter on ONE GPU
            k1 = min(args.steps, 50)
            r1 = run_single(P, S, torch, cfg_id, k1, max(min(args.warmup, 20), 5), 0.0, torch.device("cuda", 0), 0, args.preroll_ms,
                            extras=False)
            r1["steps"] = k1
            r1["note"] = "the same workload on ONE GPU (single-device step), measured in this run: the N = 1 point of this strong-scaling line"
            r1["rider"] = "one_gpu_same_workload"
            secondary.append(r1)
        sys.stdout.flush()
        _flush_c_stdio()                                      # RCCL's banner first, the JSON line last
        emit({"metric": "PHD-update steps/sec at N_particles x N_gauss x N_meas", "value": res["value"], "unit": "steps/s",
              "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "preroll_steps": res["preroll_steps"],
              "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
              "dtype": "f32", "data": "synthetic", "config": res["config"], "roofline": res["roofline"],
              "roofline_valu": res["roofline_valu"], "cpu_baseline": None, "secondary": secondary})
        return
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

    secondary = []
    if os.environ.get("PHD_BENCH_CPP_MULTI") and world == 1:
        # diagnostic: the multi-device step through the C++ host in this process, PHD_BENCH_CPP_MULTI shards on THIS GPU
        # (1 = a one-rank RCCL communicator); labelled in config, not the N = 1 headline
        torch.cuda.set_device(0)
        res = run_cpp_multi(P, S, torch, args.config or 2, args.steps, args.warmup, [0] * int(os.environ["PHD_BENCH_CPP_MULTI"]),
                            args.preroll_ms)
        sys.stdout.flush()
        _flush_c_stdio()                                      # RCCL's banner first, the JSON line last
        emit({"metric": "PHD-update steps/sec at N_particles x N_gauss x N_meas", "value": res["value"], "unit": "steps/s",
              "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "preroll_steps": res["preroll_steps"],
              "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f32", "data": "synthetic", "config": res["config"], "roofline": res["roofline"],
              "roofline_valu": res.get("roofline_valu"), "cpu_baseline": None})
        return
    if not multi:
        cfg_id = args.config or 3
        res = run_single(P, S, torch, cfg_id, args.steps, args.warmup, 0.0 if args.bare else args.cpu_seconds, dev, local_rank,
                         args.preroll_ms, extras=not args.bare, general=not args.bare)
        if not args.no_secondary and not args.bare:
            for sid in (2, 3, 5, 4, 6):
                if sid == cfg_id:
                    continue
                # riders: shorter CPU leg, no copy-ceiling / stage pass (reported once, by the headline); configs[3]
                # (16384 particles, the multi-GPU workload) on this ONE GPU is the N = 1 point of the strong-scaling curve
                k = args.steps if sid != 2 else max(args.steps, 2000)   # 23 us steps: 2000 of them are 50 ms
                if sid == 6:
                    k = min(args.steps, 40)                            # 6.5 ms steps
                r = run_single(P, S, torch, sid, k, max(args.warmup, 20 if sid != 2 else 200) if sid != 6 else 5,
                               0.0 if sid in (4, 6) else min(args.cpu_seconds, 4.0), dev, local_rank, args.preroll_ms, extras=False)
                r["steps"] = k
                r["rider"] = "cfg%d_%dx%dx%d%s" % (sid, S.CONFIGS[sid]["N"], S.CONFIGS[sid]["G"], S.CONFIGS[sid]["M"], "_cphd" if sid == 5 else "")
                secondary.append(r)
        scaling = "weak"   # N = 1: a single point of either curve; per-GPU work is what the N > 1 line divides
    else:
        cfg_id = args.config or 4
        if one_rank and not args.config:
            cfg_id = 2                                        # the dry run's historical workload: 256 x 64 x 32
        n_global = S.CONFIGS[cfg_id]["N"] if (cfg_id == 4 or one_rank) else S.CONFIGS[cfg_id]["N"] * world
        res = run_sharded(P, S, D, torch, dist, cfg_id, n_global, args.steps, args.warmup, dev, local_rank, rank, world, share,
                          one_rank, same_set=True, preroll_ms=args.preroll_ms)
        if not args.no_secondary and not args.bare and not one_rank:
            k = max(args.steps, 400)
            r = run_sharded(P, S, D, torch, dist, 2, 256 * world, k, max(args.warmup, 40), dev, local_rank, rank, world, share,
                            one_rank, same_set=False, preroll_ms=args.preroll_ms)
            r["steps"] = k
            r["shard_steps_per_s"] = r["value"] * world       # round 1's unit for this workload (ranks x steps / time)
            r["rider"] = "weak_256x64x32_per_rank"
            secondary.append(r)
            if world > 1 and cfg_id == 4:
                # the N = 1 point of THIS line, measured in this run: the same 16384-particle filter on ONE GPU (rank 0 alone,
                # fused/staged single-GPU step; the other ranks wait) — what value / N is to be compared with
                dist.barrier()
                if rank == 0:
                    k1 = min(args.steps, 50)
                    r1 = run_single(P, S, torch, 4, k1, max(min(args.warmup, 20), 5), 0.0, dev, local_rank, args.preroll_ms,
                                    extras=False)
                    r1["steps"] = k1
                    r1["note"] = ("the same workload on ONE GPU, measured by rank 0 in this run while the other ranks wait: "
                                  "the N = 1 point of this strong-scaling line")
                    r1["rider"] = "one_gpu_same_workload"
                    secondary.append(r1)
                dist.barrier()
        scaling = "strong"  # total work (one 16384-particle filter) is fixed as N grows
        # The headline of a launched N > 1 run is the C++ multi-device host as well (north_star: "host code stays C++"): rank 0
        # drives all N devices of the node through libphdslam_multi.so (ncclCommInitAll over N devices, one stream per shard,
        # one host thread) while the other ranks wait at a barrier; the one-process-per-GPU Python host measured above becomes
        # a labelled secondary.  Skipped (the Python host stays the headline, with the reason) when rank 0 cannot see N
        # devices, in the share-GPU / one-rank dry runs, or if the C++ host fails.
        if world > 1 and not share and not one_rank and cfg_id == 4:
            # the other ranks wait on the HOST (a gloo group): an RCCL barrier would keep a polling kernel resident on every
            # GPU rank 0 is about to measure
            host_group = dist.new_group(backend="gloo")
            torch.cuda.synchronize()
            dist.barrier(group=host_group)
            cpp, why = None, None
            if rank == 0:
                if torch.cuda.device_count() < world:
                    why = "rank 0 sees %d device(s), needs %d" % (torch.cuda.device_count(), world)
                else:
                    try:
                        cpp = run_cpp_multi(P, S, torch, cfg_id, args.steps, args.warmup, list(range(world)), args.preroll_ms)
                    except Exception as e:                            # the scaling record must survive: fall back, say why
                        why = "C++ host failed: %s" % (str(e)[:300],)
                    torch.cuda.set_device(local_rank)
            dist.barrier(group=host_group)
            if rank == 0:
                if cpp is not None:
                    res["note"] = ("the same sharded step driven from Python (cuda-phdslam_amd/dist.py, one process per GPU over "
                                   "torch.distributed): secondary to the C++ host of this line")
                    res["steps"] = args.steps
                    res["rider"] = "python_host_one_process_per_gpu"
                    secondary.append(res)
                    res = cpp
                else:
                    res["config"]["cpp_multi_host"] = False
                    res["config"]["cpp_multi_host_skipped"] = why

    if multi:
        # RCCL prints a version banner through C stdio at communicator creation; on a pipe it stays in the C buffer until the
        # process exits — AFTER Python's own output.  Flush it on every rank now, then print: the JSON line is the last line.
        sys.stdout.flush()
        _flush_c_stdio()
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "PHD-update steps/sec at N_particles x N_gauss x N_meas",
            "value": res["value"],
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_steps": res["preroll_steps"],
            "ms_per_step": res["ms_per_step"],
            "ms_per_step_gpu_p10_p50_p90": res.get("ms_per_step_gpu_p10_p50_p90"),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": res["config"],
            "stages_us_per_workgroup": res.get("stages_us_per_workgroup"),
            "roofline": res["roofline"],
            "roofline_valu": res["roofline_valu"],
            "cpu_baseline": res.get("cpu_baseline"),
            "value_general": res.get("value_general"),
            "secondary": secondary,
        }
        emit(out)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
