// phdslam — command-line driver with the reference executable's contract (src/main.cpp:1442-1500,
// run_synth :1075-1322): argv[1] = config file in the reference's cfg/config.cfg format,
// argv[2] = "synth".  Reads <data_directory>/measurements.txt and controls.txt (optionally
// measurement_times.txt / control_times.txt for asynchronous inputs, traj.txt with
// follow_trajectory), runs the Rao-Blackwellised GM-PHD-SLAM filter on the GPU with the state
// resident on the device, and writes state_estimate%05d.log (README:31-39 5-line format, or HEAD's
// 7-line writeLog with --log7) and loopTime.log.
//
//   phdslam <config.cfg> [synth] [--out DIR] [--seed S] [--capacity C] [--steps K] [--log7] [--device-noise] [--devices N [--shards S]]
//
// environment: PHD_DRIVER_SYNC=1 the step-synchronous loop (run_synth's own structure) instead of the pipelined one;
// PHD_DRIVER_PROFILE=1 per-phase times of the synchronous loop; PHD_DRIVER_SYNC_LOG=1 logs written in line.
//
// --devices N: ONE filter sharded over N GPUs of this process (include/phdslam_multi.h: one shard and one stream per
// device, RCCL all-gather of the log-weights, global resample, particle migration over xGMI).  --shards S > N puts several
// shards on a device (exchange by device copies; for boxes with fewer GPUs than shards).  Same files in, same logs out.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <condition_variable>
#include <algorithm>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "phdfilter_compat.h"
#include "phdslam_multi.h"

static void die(const char* where)
{
    fprintf(stderr, "%s failed: %s\n", where, phd_last_error());
    exit(EXIT_FAILURE);
}
#define CHK(call) do { if ((call) != PHD_OK) die(#call); } while (0)

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0); // per-step progress lines reach a pipe as they are printed
    if (argc < 2) {
        fprintf(stderr, "usage: %s <config.cfg> [synth] [--out DIR] [--seed S] [--capacity C] [--steps K] [--log7] [--device-noise] [--devices N [--shards S]]\n", argv[0]);
        return 2;
    }
    std::string out_dir = ".";
    uint64_t seed = 1;
    int capacity = 0, max_steps = -1, n_devices = 0, n_shards = 0;
    bool log7 = false;
    // --device-noise: the control noise of the vehicle predict is drawn on the device (counter-based generator seeded with
    // --seed, one draw per predicted particle by its global index) instead of 2 N randn() calls on the host per step as the
    // reference does (src/phdfilter.cu:1147-1152) — a quarter of the loop time at 4096 particles.  Another random stream,
    // the same distribution (the reference's own stream is seeded from the wall clock: there is nothing to reproduce).
    bool device_noise = false;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "synth")) continue;
        if (!strcmp(argv[i], "disparity")) { fprintf(stderr, "the disparity pipeline is out of scope\n"); return 2; }
        if (!strcmp(argv[i], "--log7")) log7 = true;
        else if (!strcmp(argv[i], "--device-noise")) device_noise = true;
        else if (i + 1 < argc && !strcmp(argv[i], "--out")) out_dir = argv[++i];
        else if (i + 1 < argc && !strcmp(argv[i], "--seed")) seed = strtoull(argv[++i], nullptr, 10);
        else if (i + 1 < argc && !strcmp(argv[i], "--capacity")) capacity = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--steps")) max_steps = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--devices")) n_devices = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--shards")) n_shards = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    char data_dir[4096];
    int32_t n_steps_cfg = -1;
    CHK(phd_config_load(argv[1], &config, data_dir, sizeof(data_dir), &n_steps_cfg));       // main.cpp:1463
    phd_compat_seed_rng(seed);

    // inputs (main.cpp:1078-1093): data_directory must end with '/' (string concatenation, :1079)
    const std::string dd(data_dir);
    const std::string mfile = dd + "measurements.txt", cfile = dd + "controls.txt";
    size_t n_meas_steps = 0, n_total = 0, n_ctrl = 0, n_mt = 0, n_ct = 0, n_traj = 0;
    CHK(phd_load_measurements(mfile.c_str(), 0, nullptr, 0, nullptr, 0, &n_meas_steps, &n_total));
    std::vector<phd_measurement> meas(n_total ? n_total : 1);
    std::vector<int32_t> sizes(n_meas_steps ? n_meas_steps : 1);
    CHK(phd_load_measurements(mfile.c_str(), 0, meas.data(), meas.size(), sizes.data(), sizes.size(), &n_meas_steps, &n_total));
    std::vector<size_t> moff(n_meas_steps + 1, 0);
    for (size_t k = 0; k < n_meas_steps; ++k) moff[k + 1] = moff[k] + sizes[k];
    CHK(phd_load_controls(cfile.c_str(), -1, nullptr, 0, &n_ctrl));
    std::vector<phd_ackerman_control> controls(n_ctrl ? n_ctrl : 1);
    CHK(phd_load_controls(cfile.c_str(), -1, controls.data(), controls.size(), &n_ctrl));
    CHK(phd_load_timestamps((dd + "measurement_times.txt").c_str(), nullptr, 0, &n_mt));
    CHK(phd_load_timestamps((dd + "control_times.txt").c_str(), nullptr, 0, &n_ct));
    std::vector<float> mtimes(n_mt ? n_mt : 1), ctimes(n_ct ? n_ct : 1);
    CHK(phd_load_timestamps((dd + "measurement_times.txt").c_str(), mtimes.data(), mtimes.size(), &n_mt));
    CHK(phd_load_timestamps((dd + "control_times.txt").c_str(), ctimes.data(), ctimes.size(), &n_ct));
    const bool has_timestamps = n_mt > 0;                                                     // :1094
    printf("Loaded %zu measurement steps, %zu control inputs, %zu/%zu time stamps\n", n_meas_steps, n_ctrl, n_mt, n_ct);
    int nSteps;
    if (!has_timestamps) {
        nSteps = (int)n_meas_steps;                                                          // :1097-1099
    } else {
        if (n_mt != n_meas_steps) { fprintf(stderr, "mismatched measurements and measurement timestamps!\n"); return 1; } // :1103-1107
        if (n_ct != n_ctrl) { fprintf(stderr, "mismatched controls and controls timestamps!\n"); return 1; }              // :1108-1112
        nSteps = (int)(n_mt + n_ct);                                                         // :1115
    }
    if (n_steps_cfg > 0 && nSteps > n_steps_cfg) nSteps = n_steps_cfg;                       // :1117-1118
    if (max_steps > 0 && nSteps > max_steps) nSteps = max_steps;

    std::vector<phd_pose> traj;
    if (config.followTrajectory) {                                                           // :1121-1127
        CHK(phd_load_trajectory((dd + "traj.txt").c_str(), nullptr, 0, &n_traj));
        traj.resize(n_traj ? n_traj : 1);
        CHK(phd_load_trajectory((dd + "traj.txt").c_str(), traj.data(), traj.size(), &n_traj));
        config.n_particles = 1;
        config.nPredictParticles = 1;
    }

    int max_m = 1;
    for (size_t k = 0; k < n_meas_steps; ++k) max_m = sizes[k] > max_m ? sizes[k] : max_m;
    if (n_devices > 0 || n_shards > 0) {
        // ---- the sharded filter (run_synth's loop, src/main.cpp:1178-1312, over include/phdslam_multi.h) ----
        if (config.followTrajectory || config.filterType == 1 || log7) {
            fprintf(stderr, "--devices: follow_trajectory, filter_type = 1 and --log7 run on a single device\n");
            return 2;
        }
        if (n_devices <= 0) n_devices = 1;
        if (n_shards <= 0) n_shards = n_devices;
        std::vector<int32_t> devs((size_t)n_shards);
        for (int k = 0; k < n_shards; ++k) devs[k] = k % n_devices;
        phd_multi_options mo = {};
        mo.n_shards = n_shards;
        mo.devices = devs.data();
        mo.map_capacity = capacity > 0 ? capacity : 512;
        mo.max_measurements = max_m < PHD_MAX_MEASUREMENTS ? max_m : PHD_MAX_MEASUREMENTS;
        phd_multi* m = nullptr;
        CHK(phd_multi_create(&config, &mo, &m));
        if (device_noise) CHK(phd_multi_seed(m, seed));
        const int N0 = config.n_particles;
        // particle shotgun (n_predict_particles = k, src/phdfilter.cu:1185-1238): the set grows k-fold per predict, up to 5 n k
        const int kshot_m = config.nPredictParticles > 1 ? config.nPredictParticles : 1;
        const size_t N_max = (size_t)N0 * (kshot_m > 1 ? 5 * kshot_m : 1);
        const int ex = phd_multi_exchange(m);
        printf("sharded filter: %d particles over %d shards on %d device(s), transport %s, exchange %s\n", N0, n_shards, n_devices,
               phd_multi_uses_rccl(m) ? "RCCL" : "device copies",
               ex == PHD_EXCHANGE_GATHERED ? "whole-shard all-gather" : ex == PHD_EXCHANGE_PULL ? "direct reads of the owners' slabs" : "all-to-all");
        std::vector<phd_pose> poses(N_max);
        std::vector<float> logw(N_max);
        std::vector<phd_gaussian2d> map((size_t)mo.map_capacity), eap(4 * (size_t)mo.map_capacity);
        std::vector<phd_ackerman_noise> noise(N_max);
        const std::string timefile = out_dir + "/loopTime.log";
        printf("STARTING SIMULATION\n");
        size_t z_idx = 0, c_idx = 0;
        float last_time = 0, current_time = 0;
        phd_ackerman_control current_control = {0, 0};
        for (int n = 0; n < nSteps; ++n) {
            timeval t0, t1;
            gettimeofday(&t0, nullptr);
            const phd_measurement* Z = nullptr;
            int M = 0;
            if (has_timestamps) {                                                            // :1189-1229
                if (z_idx >= n_mt || c_idx >= n_ct) { printf("no more timestamps\n"); break; }
                const bool meas_first = mtimes[z_idx] < ctimes[c_idx], both = mtimes[z_idx] == ctimes[c_idx];
                last_time = current_time;
                current_time = ctimes[c_idx];
                config.dt = current_time - last_time;
                CHK(phd_multi_set_config(m, &config));
                if (meas_first) { Z = meas.data() + moff[z_idx]; M = sizes[z_idx]; z_idx++; }
                else if (both) { current_control = controls[c_idx++]; Z = meas.data() + moff[z_idx]; M = sizes[z_idx]; z_idx++; }
                else current_control = controls[c_idx++];
            } else {                                                                         // :1231-1237 lock-step
                Z = meas.data() + moff[n]; M = sizes[n];
                if (n > 0) {
                    if ((size_t)(n - 1) >= n_ctrl) { fprintf(stderr, "not enough controls\n"); break; }
                    current_control = controls[n - 1];
                }
            }
            if (n > 0) {                                                                     // no motion at step 0 (:1244)
                // (subdivide_predict > 1: the extra predicts carry no scan)
                const int sub = config.subdividePredict > 0 ? config.subdividePredict : 1;
                for (int s = 0; s < sub; ++s) {
                    const int n_pred = phd_multi_n_particles_now(m) * kshot_m;               // one draw per PREDICTED particle
                    if ((size_t)n_pred > N_max) { fprintf(stderr, "particle count would exceed 5 n_particles n_predict_particles\n"); return 1; }
                    for (int i = 0; i < n_pred && !device_noise; ++i) {                      // phdfilter.cu:1147-1152
                        noise[i].n_alpha = (float)(config.stdAlpha * randn());
                        noise[i].n_encoder = (float)(config.stdEncoder * randn());
                    }
                    const bool last = s == sub - 1;
                    CHK(phd_multi_update(m, &current_control, device_noise ? nullptr : noise.data(), last ? Z : nullptr, last ? M : 0));
                }
            } else {
                CHK(phd_multi_update(m, nullptr, nullptr, Z, M));                            // step 0: the scan alone
            }
            phd_pose expected;
            int32_t n_map = 0, who = 0;
            phd_step_report rep;
            {
                const int rc = phd_multi_state_snapshot(m, &expected, map.data(), (int)map.size(), &n_map, &who, poses.data(), logw.data(), &rep);
                if (rc == PHD_ERR_NAN) { printf("nan weights detected! exiting...\n"); break; }
                if (rc != PHD_OK) die("phd_multi_state_snapshot");
            }
            const int N = phd_multi_n_particles_now(m);
            if ((config.mapEstimate & 2) && N > 1) {                                         // :363-379
                int32_t n_eap = 0;
                int rc = phd_multi_expected_map(m, eap.data(), (int)eap.size(), &n_eap);
                if (rc == PHD_ERR_CAPACITY && n_eap > (int)eap.size()) { eap.resize((size_t)n_eap); rc = phd_multi_expected_map(m, eap.data(), (int)eap.size(), &n_eap); }
                if (rc != PHD_OK) die("phd_multi_expected_map");
                if (!(config.mapEstimate & 1)) {
                    if ((size_t)n_eap > map.size()) map.resize((size_t)n_eap);
                    std::copy(eap.begin(), eap.begin() + n_eap, map.begin());
                    n_map = n_eap;
                }
            }
            const double u = randu01();
            const int did = ((M > 0 && rep.neff <= config.resampleThresh) || N > 5 * N0) ? 1 : 0;   // :1286
            if (did) CHK(phd_multi_resample(m, u));
            CHK(phd_write_state_log(out_dir.c_str(), n, &expected, map.data(), n_map, logw.data(), poses.data(), N, config.maxCardinality));
            gettimeofday(&t1, nullptr);
            const double elapsed = (t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_usec - t0.tv_usec) / 1000.0;
            if (FILE* tf = fopen(timefile.c_str(), "a")) { fprintf(tf, "%g\n", elapsed); fclose(tf); }
            printf("****** Time Step [%d/%d] ****** M=%d particles=%d map=%d resampled=%d %.3f ms\n", n, nSteps, M, N, n_map, did, elapsed);
        }
        phd_multi_destroy(m);
        return 0;
    }
    phd_options opt = {};
    opt.n_particles = config.n_particles;
    opt.map_capacity = capacity > 0 ? capacity : 512;
    opt.max_measurements = max_m < PHD_MAX_MEASUREMENTS ? max_m : PHD_MAX_MEASUREMENTS;
    phd_filter* f = nullptr;
    CHK(phd_create(&config, &opt, &f));   // particles start at the configured pose, weights -log N (:1130-1145)
    if (device_noise) CHK(phd_seed(f, seed));
    const int kshot = config.nPredictParticles > 1 ? config.nPredictParticles : 1;
    const int n_max = config.n_particles * (kshot > 1 ? 5 * kshot : 1);

    // The state log of a step (every particle's pose and weight as text: half the loop time at 256 particles, most of it at
    // 4096) is formatted and written by a few writer threads (one file per step, so they are independent) while the filter
    // runs the next steps; at most eight steps are in flight.  PHD_DRIVER_SYNC_LOG=1 writes in line, as the reference does.
    struct LogJob {
        int step, n_map, n_cur;
        bool did;
        phd_pose expected;
        std::vector<phd_gaussian2d> map;
        std::vector<float> logw, cn;
        std::vector<phd_pose> poses;
        std::vector<int32_t> ridx;
    };
    const bool cphd_log = config.filterType == 1;
    const int max_card = config.maxCardinality;
    auto write_job = [&](const LogJob& j) -> int {
        if (log7) {
            if (cphd_log)
                return phd_write_state_log7_cphd(out_dir.c_str(), j.step, &j.expected, j.map.data(), j.n_map, j.logw.data(),
                                                 j.poses.data(), j.ridx.data(), j.n_cur, kshot, j.cn.data(), (int)j.cn.size());
            return phd_write_state_log7(out_dir.c_str(), j.step, &j.expected, j.map.data(), j.n_map, j.logw.data(), j.poses.data(),
                                        j.ridx.data(), j.n_cur, max_card, kshot);
        }
        if (cphd_log)
            return phd_write_state_log_cphd(out_dir.c_str(), j.step, &j.expected, j.map.data(), j.n_map, j.logw.data(),
                                            j.poses.data(), j.n_cur, j.cn.data(), (int)j.cn.size());
        return phd_write_state_log(out_dir.c_str(), j.step, &j.expected, j.map.data(), j.n_map, j.logw.data(), j.poses.data(),
                                   j.n_cur, max_card);
    };
    const bool sync_log = getenv("PHD_DRIVER_SYNC_LOG") != nullptr;
    std::mutex log_mu;
    std::condition_variable log_cv;
    std::deque<LogJob> log_q;
    bool log_done = false;
    int log_err = 0, log_inflight = 0;
    std::vector<std::thread> log_threads;
    if (!sync_log)
        // (writer threads: a 4096-particle log is ~29 000 numbers; the filter produces two of them per millisecond)
        for (unsigned t = 0, nt = std::max(2u, std::min(8u, std::thread::hardware_concurrency() / 2)); t < nt; ++t)
            log_threads.emplace_back([&]() {
                for (;;) {
                    LogJob j;
                    {
                        std::unique_lock<std::mutex> lk(log_mu);
                        log_cv.wait(lk, [&] { return log_done || !log_q.empty(); });
                        if (log_q.empty()) return;
                        j = std::move(log_q.front());
                        log_q.pop_front();
                        ++log_inflight;
                    }
                    const int rc = write_job(j);
                    {
                        std::lock_guard<std::mutex> lk(log_mu);
                        --log_inflight;
                        if (rc && !log_err) log_err = rc;
                    }
                    log_cv.notify_all();
                }
            });
    auto finish_logs = [&]() {
        if (sync_log) return;
        { std::lock_guard<std::mutex> lk(log_mu); log_done = true; }
        log_cv.notify_all();
        for (auto& th : log_threads) if (th.joinable()) th.join();
    };

    std::vector<phd_pose> poses(n_max);
    std::vector<float> logw(n_max);
    std::vector<int32_t> ridx(n_max);
    std::vector<phd_gaussian2d> map(opt.map_capacity), eap(4 * (size_t)opt.map_capacity);
    std::vector<phd_ackerman_noise> noise((size_t)n_max);
    std::vector<float> cn_est;
    std::string timefile = out_dir + "/loopTime.log";
    printf("STARTING SIMULATION\n");
    size_t z_idx = 0, c_idx = 0;
    float last_time = 0, current_time = 0;
    phd_ackerman_control current_control = {0, 0};                                           // :1170-1172
    // PHD_DRIVER_PROFILE=1: where the loop time goes (inputs + predict, update, state extraction, resample, log writing)
    const bool prof = getenv("PHD_DRIVER_PROFILE") != nullptr;
    double acc_t[5] = {0, 0, 0, 0, 0};
    auto now = []() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; };
    double tp = 0;
    // per step: the five phases, and what the step was made of — <out>/loopProfile.log, one line per step:
    //   step M map_size resampled | inputs+predict update state_extraction resample log_hand_off  (ms)
    // (tools/e2e_run.py turns it into per-phase percentiles and names the phase the slow steps spend their time in)
    double step_t[5] = {0, 0, 0, 0, 0};
    FILE* prof_file = prof ? fopen((out_dir + "/loopProfile.log").c_str(), "w") : nullptr;
#define PROF_MARK(k) do { if (prof) { const double t_ = now(); acc_t[k] += t_ - tp; step_t[k] = t_ - tp; tp = t_; } } while (0)
    // ---- the pipelined loop (round 6): the plain configuration — one predicted particle per prior particle, MAP map, PHD or CPHD filter —
    // runs WITHOUT a host synchronisation inside the step.  run_synth's step is predict -> update -> recoverSlamState -> log ->
    // nEff test -> resample (src/main.cpp:1244-1297), with the host waiting for the device three times; here the host only
    // enqueues: the control noise of step n + 1 and its resampling uniform are drawn by a helper thread (the SAME random stream in
    // the SAME order: noise of the step, then its uniform — the draws do not depend on the filter's state), the state of step n is
    // captured on the device between its update and its resample (the order of the log's contents, :1271-1297), the resample is
    // decided on the device from the same nEff float, and the log of step n is handed to the writers while step n + 1 runs.
    // Same files, character for character (tests/test_gpu_driver.py compares the two loops); PHD_DRIVER_SYNC=1 runs the loop below.
    const bool pipelined = !config.followTrajectory && kshot == 1 && !config.savePrediction && !(config.mapEstimate & 2) &&
                           getenv("PHD_DRIVER_SYNC") == nullptr;
    if (pipelined) {
        const int n_part = phd_n_particles(f);
        const int sub = config.subdividePredict > 0 ? config.subdividePredict : 1;
        const size_t slot_entries = (size_t)n_part * sub;
        phd_ackerman_noise* ring[3] = {nullptr, nullptr, nullptr};
        for (auto& r : ring) { r = (phd_ackerman_noise*)phd_host_alloc(slot_entries * sizeof(phd_ackerman_noise)); if (!r) die("phd_host_alloc"); }
        std::vector<double> uniforms((size_t)nSteps > 0 ? nSteps : 1);
        std::mutex mu;
        std::condition_variable cv;
        int produced = 0, retired = -1;                      // steps whose draws are ready; last step whose snapshot was waited for
        bool stop = false;
        std::thread producer([&]() {
            for (int m = 0; m < nSteps; ++m) {
                {   // slot m % 3 is free once step m - 3 has been retired (its upload is behind the device by then)
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || retired >= m - 3; });
                    if (stop) return;
                }
                if (m > 0 && !device_noise) {
                    phd_ackerman_noise* nz = ring[m % 3];
                    for (size_t i = 0; i < slot_entries; ++i) {                              // phdfilter.cu:1147-1152
                        nz[i].n_alpha = (float)(config.stdAlpha * randn());
                        nz[i].n_encoder = (float)(config.stdEncoder * randn());
                    }
                }
                uniforms[m] = randu01();
                { std::lock_guard<std::mutex> lk(mu); produced = m + 1; }
                cv.notify_all();
            }
        });
        struct Pending { int M; bool resample_enqueued; double elapsed; int inst; };
        std::vector<Pending> pend((size_t)nSteps > 0 ? nSteps : 1);
        bool failed_nan = false;
        int last_n_map = 0, last_did = 0;
        FILE* time_file = fopen(timefile.c_str(), "a");
        if (prof_file) fprintf(prof_file, "# pipelined: wait_for_draws enqueue_step wait_for_device log_hand_off time_file+print\n");
        // retire step k: wait for its download, hand its log to the writers, print its line
        auto retire = [&](int k) -> bool {
            phd_snapshot_view v;
            const int rc = phd_snapshot_wait(f, k & 1, &v);
            PROF_MARK(2);                                                                        // waiting for the device (step k's download)
            if (rc == PHD_ERR_NAN) { printf("nan weights detected! exiting...\n"); failed_nan = true; return false; }      // :1307-1311
            if (rc != PHD_OK) die("phd_snapshot_wait");
            const int n_cur = v.n_particles;
            const int did = pend[k].resample_enqueued ? v.report.did_resample : 0;
            LogJob j;
            j.step = k; j.n_map = v.n_map; j.n_cur = n_cur; j.did = did != 0; j.expected = *v.expected;
            j.map.assign(v.map, v.map + v.n_map);
            j.logw.assign(v.log_weights, v.log_weights + n_cur);
            j.poses.assign(v.poses, v.poses + n_cur);
            if (cphd_log && v.cardinality) j.cn.assign(v.cardinality, v.cardinality + v.cardinality_len);   // cn_estimate (:360)
            if (log7) {
                j.ridx.resize((size_t)n_cur);
                if (did && v.resample_idx) std::copy(v.resample_idx, v.resample_idx + n_cur, j.ridx.begin());
                else for (int i = 0; i < n_cur; ++i) j.ridx[i] = i;                                               // :1292-1296
            }
            if (sync_log) {
                CHK(write_job(j));
            } else {
                std::unique_lock<std::mutex> lk(log_mu);
                log_cv.wait(lk, [&] { return (int)log_q.size() + log_inflight < 8; });
                if (log_err) { lk.unlock(); finish_logs(); die("state log"); }
                log_q.push_back(std::move(j));
                lk.unlock();
                log_cv.notify_all();
            }
            PROF_MARK(3);                                                                        // log hand-off
            last_n_map = v.n_map; last_did = did;
            // loopTime.log (:1300-1305: appended every step; here the file stays open and is flushed every step — the same content, a
            // few microseconds of a 40 us step at the reference's default 200 particles less)
            if (time_file) { fprintf(time_file, "%g\n", pend[k].elapsed); fflush(time_file); }
            printf("****** Time Step [%d/%d] ****** M=%d particles=%d map=%d resampled=%d inst=%d %.3f ms\n", k, nSteps, pend[k].M, n_cur, v.n_map, did,
                   pend[k].inst, pend[k].elapsed);
            { std::lock_guard<std::mutex> lk(mu); retired = k; }
            cv.notify_all();
            return true;
        };
        int enqueued = 0;
        for (int n = 0; n < nSteps; ++n) {
            timeval t0, t1;
            gettimeofday(&t0, nullptr);
            const phd_measurement* Z = nullptr;
            int M = 0;
            if (has_timestamps) {                                                                // :1189-1229
                if (z_idx >= n_mt || c_idx >= n_ct) { printf("no more timestamps\n"); break; }
                const bool meas_first = mtimes[z_idx] < ctimes[c_idx], both = mtimes[z_idx] == ctimes[c_idx];
                last_time = current_time;
                current_time = ctimes[c_idx];      // (sic) the reference takes the control's time stamp in all three branches
                config.dt = current_time - last_time;
                setDeviceConfig(config);
                CHK(phd_set_config(f, &config));
                if (meas_first) { Z = meas.data() + moff[z_idx]; M = sizes[z_idx]; z_idx++; }
                else if (both) { current_control = controls[c_idx++]; Z = meas.data() + moff[z_idx]; M = sizes[z_idx]; z_idx++; }
                else current_control = controls[c_idx++];
            } else {                                                                             // :1231-1237 lock-step
                Z = meas.data() + moff[n]; M = sizes[n];
                if (n > 0) {
                    if ((size_t)(n - 1) >= n_ctrl) { fprintf(stderr, "not enough controls\n"); break; }
                    current_control = controls[n - 1];
                }
            }
            if (prof) { tp = now(); for (double& v : step_t) v = 0; }
            {   // this step's draws
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return produced > n; });
            }
            PROF_MARK(0);                                                                        // waiting for the helper thread's draws
            // predict (no motion at step 0, :1244) and update (:1260-1272): the step's last predict and its update are ONE launch
            // (phd_predict_update: the same results as the two calls)
            if (n > 0) {
                for (int s = 0; s + 1 < sub; ++s)
                    CHK(phd_predict_ackerman(f, current_control, device_noise ? nullptr : ring[n % 3] + (size_t)s * n_part));
                CHK(phd_predict_update(f, current_control, device_noise ? nullptr : ring[n % 3] + (size_t)(sub - 1) * n_part, Z, M));
            } else if (M > 0) {
                CHK(phd_update(f, Z, M));
            }
            pend[n].inst = M > 0 ? phd_debug_update_instantiation(f) : -1;
            CHK(phd_snapshot_capture(f, n & 1));                                                 // the log's contents: before the resample
            // nEff test and resampling (:1281-1297), decided on the device: nEff <= resample_threshold and the step had a scan
            if (M > 0) CHK(phd_resample_if_needed(f, uniforms[n], 1, nullptr, nullptr));
            CHK(phd_snapshot_send(f, n & 1, log7 && M > 0));
            PROF_MARK(1);                                                                        // enqueueing the whole step
            pend[n].M = M;
            pend[n].resample_enqueued = M > 0;
            enqueued = n + 1;
            bool ok = true;
            if (n > 0) ok = retire(n - 1);
            PROF_MARK(4);                                                                        // (retire() marks 2 and 3 inside)
            if (prof_file && n > 0) fprintf(prof_file, "%d %d %d %d %.4f %.4f %.4f %.4f %.4f\n", n, M, last_n_map, last_did, step_t[0], step_t[1],
                                            step_t[2], step_t[3], step_t[4]);
            gettimeofday(&t1, nullptr);
            pend[n].elapsed = (t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_usec - t0.tv_usec) / 1000.0;
            if (!ok) break;
        }
        if (!failed_nan && enqueued > 0) retire(enqueued - 1);
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        producer.join();
        if (time_file) fclose(time_file);
        phd_sync(f);
        for (auto& r : ring) phd_host_free(r);
    }
    for (int n = 0; n < (pipelined ? 0 : nSteps); ++n) {
        timeval t0, t1;
        gettimeofday(&t0, nullptr);
        if (prof) { tp = now(); for (double& v : step_t) v = 0; }   // a phase this step does not reach reads 0, not the last step's value
        // inputs of this step (:1187-1237)
        const phd_measurement* Z = nullptr;
        int M = 0;
        bool do_predict = false;
        if (has_timestamps) {
            if (z_idx >= n_mt || c_idx >= n_ct) { printf("no more timestamps\n"); break; }  // :1189-1192
            const bool meas_first = mtimes[z_idx] < ctimes[c_idx], both = mtimes[z_idx] == ctimes[c_idx];
            last_time = current_time;
            current_time = ctimes[c_idx];          // (sic) the reference takes the control's time stamp in all three branches
            config.dt = current_time - last_time;
            setDeviceConfig(config);
            CHK(phd_set_config(f, &config));
            if (meas_first) {                                                                // :1193-1203 measurement only
                Z = meas.data() + moff[z_idx]; M = sizes[z_idx]; z_idx++;
            } else if (both) {                                                               // :1204-1215 odometry and measurement
                current_control = controls[c_idx++];
                Z = meas.data() + moff[z_idx]; M = sizes[z_idx]; z_idx++;
            } else {                                                                         // :1216-1229 odometry only
                current_control = controls[c_idx++];
            }
            do_predict = true;
        } else {                                                                             // :1231-1237 lock-step
            Z = meas.data() + moff[n]; M = sizes[n];
            if (n > 0) {
                if ((size_t)(n - 1) >= n_ctrl) { fprintf(stderr, "not enough controls\n"); break; }
                current_control = controls[n - 1];
            }
            do_predict = true;
        }
        int n_cur = phd_n_particles(f);
        if (config.followTrajectory) {                                                       // :1239-1243
            const phd_pose q = traj[(size_t)n < n_traj ? n : n_traj - 1];
            float lw0 = 0.f;
            CHK(phd_set_particles(f, &q, &lw0, 1));
        } else if (n > 0 && do_predict) {                                                    // no motion at step 0 (:1244)
            for (int s = 0; s < (config.subdividePredict > 0 ? config.subdividePredict : 1); ++s) {
                n_cur = phd_n_particles(f);
                for (int i = 0; i < n_cur * kshot && !device_noise; ++i) {                   // phdfilter.cu:1147-1152
                    noise[i].n_alpha = (float)(config.stdAlpha * randn());
                    noise[i].n_encoder = (float)(config.stdEncoder * randn());
                }
                CHK(phd_predict_ackerman(f, current_control, device_noise ? nullptr : noise.data()));
            }
        }
        if (config.savePrediction) {
            // save_prediction (:1256-1257): the reference dumps the predicted particle set to a .mat file
            // (libmatio); here the same state as text — log-weights, then poses (maps are untouched by predict)
            n_cur = phd_n_particles(f);
            CHK(phd_get_particles(f, poses.data(), logw.data()));
            char name[64];
            snprintf(name, sizeof name, "/particles_predict%05d.log", n);
            if (FILE* pf = fopen((out_dir + name).c_str(), "w")) {
                for (int i = 0; i < n_cur; ++i) fprintf(pf, "%g ", logw[i]);
                fprintf(pf, "\n");
                for (int i = 0; i < n_cur; ++i)
                    fprintf(pf, "%g %g %g %g %g %g ", poses[i].px, poses[i].py, poses[i].ptheta, poses[i].vx, poses[i].vy,
                            poses[i].vtheta);
                fprintf(pf, "\n");
                fclose(pf);
            }
        }
        PROF_MARK(0);
        if (M > 0) CHK(phd_update(f, Z, M));                                                 // :1260-1272
        if (prof) { phd_sync(f); }
        PROF_MARK(1);
        // state extraction (:1274) and log
        n_cur = phd_n_particles(f);
        phd_pose expected;
        int32_t n_map = 0, who = 0;
        // weighted-mean pose, arg-max particle's map, all poses and weights AND the step's report (capacity status, the
        // nEff the update's weights routine computed): one call, the ONE host synchronisation of the step
        phd_step_report rep;
        {
            const int rc = phd_state_snapshot(f, &expected, map.data(), (int)map.size(), &n_map, &who, poses.data(), logw.data(), &rep);
            if (rc == PHD_ERR_NAN) { printf("nan weights detected! exiting...\n"); break; }              // :1307-1311
            if (rc != PHD_OK) die("phd_state_snapshot");
        }
        if ((config.mapEstimate & 2) && n_cur > 1) {
            // expected-a-posteriori map (recoverSlamState, :363-379).  map_estimate = 2: it is the map of the log;
            // map_estimate = 3 (both): the MAP map stays in the log, the EAP map goes to expected_mapNNNNN.log
            int32_t n_eap = 0;
            int rc = phd_expected_map(f, eap.data(), (int)eap.size(), &n_eap);
            if (rc == PHD_ERR_CAPACITY && n_eap > (int)eap.size()) {
                eap.resize((size_t)n_eap);
                rc = phd_expected_map(f, eap.data(), (int)eap.size(), &n_eap);
            }
            if (rc != PHD_OK) die("phd_expected_map");
            if (config.mapEstimate & 1) {
                char name[64];
                snprintf(name, sizeof name, "/expected_map%05d.log", n);
                if (FILE* ef = fopen((out_dir + name).c_str(), "w")) {
                    for (int i = 0; i < n_eap; ++i)
                        fprintf(ef, "%g %g %g %g %g %g %g ", eap[i].weight, eap[i].mean[0], eap[i].mean[1], eap[i].cov[0],
                                eap[i].cov[1], eap[i].cov[2], eap[i].cov[3]);
                    fprintf(ef, "\n");
                    fclose(ef);
                }
            } else {
                if ((size_t)n_eap > map.size()) map.resize((size_t)n_eap);
                std::copy(eap.begin(), eap.begin() + n_eap, map.begin());
                n_map = n_eap;
            }
        }
        if (config.filterType == 1) {                                                        // cn_estimate (:360)
            cn_est.resize((size_t)phd_cardinality_length(f));
            CHK(phd_cardinality_estimate(f, cn_est.data(), nullptr));
        }
        PROF_MARK(2);
        // nEff test and resampling (:1281-1297)
        // The trigger (:1286) is decided here from the report's nEff — the same float the device would compare — so the
        // resample is enqueued without waiting for anything (the parent indices come back only for the 7-line log).
        // A grown particle set (shotgun) keeps the library's own trigger, which also tests N > 5 n_particles.
        int32_t did = 0;
        {
            const double u = randu01();
            if (kshot > 1) {
                CHK(phd_resample_if_needed(f, u, M > 0, &did, ridx.data()));
            } else {
                did = (M > 0 && rep.neff <= config.resampleThresh) ? 1 : 0;
                if (did) CHK(phd_resample(f, &u, 1, log7 ? ridx.data() : nullptr));
                else if (log7) for (int i = 0; i < n_cur; ++i) ridx[i] = i;                              // :1292-1296
            }
        }
        PROF_MARK(3);
        {
            // weights and poses are those of the step (before resampling); a resample that shrinks a grown
            // (shotgun) particle set yields fewer parent indices than particles: the rest is marked -1
            if (log7 && did) for (int i = phd_n_particles(f); i < n_cur; ++i) ridx[i] = -1;
            LogJob j;
            j.step = n; j.n_map = n_map; j.n_cur = n_cur; j.did = did != 0; j.expected = expected;
            j.map.assign(map.begin(), map.begin() + n_map);
            j.logw.assign(logw.begin(), logw.begin() + n_cur);
            j.poses.assign(poses.begin(), poses.begin() + n_cur);
            if (log7) j.ridx.assign(ridx.begin(), ridx.begin() + n_cur);
            if (cphd_log) j.cn = cn_est;
            if (sync_log) {
                CHK(write_job(j));
            } else {
                std::unique_lock<std::mutex> lk(log_mu);
                log_cv.wait(lk, [&] { return (int)log_q.size() + log_inflight < 8; });
                if (log_err) { lk.unlock(); finish_logs(); die("state log"); }
                log_q.push_back(std::move(j));
                lk.unlock();
                log_cv.notify_all();
            }
        }
        PROF_MARK(4);
        gettimeofday(&t1, nullptr);
        double elapsed = (t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_usec - t0.tv_usec) / 1000.0;
        if (FILE* tf = fopen(timefile.c_str(), "a")) { fprintf(tf, "%g\n", elapsed); fclose(tf); } // :1300-1305
        // (inst: which instantiation of the update kernel the step's scan ran, csrc/phd_kernels.hip; -1: no scan this step)
        printf("****** Time Step [%d/%d] ****** M=%d particles=%d map=%d resampled=%d inst=%d %.3f ms\n", n, nSteps, M, n_cur, n_map, did,
               M > 0 ? phd_debug_update_instantiation(f) : -1, elapsed);
        if (prof_file) fprintf(prof_file, "%d %d %d %d %.4f %.4f %.4f %.4f %.4f\n", n, M, n_map, (int)did, step_t[0], step_t[1], step_t[2],
                               step_t[3], step_t[4]);
    }
    if (prof_file) fclose(prof_file);
    if (prof && nSteps > 0 && pipelined)
        printf("loop profile, pipelined loop (host ms per step): waiting for the helper thread's draws %.3f, enqueueing the step %.3f, waiting for the device "
               "(the previous step's download) %.3f, log hand-off %.3f, time file + progress line %.3f\n",
               acc_t[0] / nSteps, acc_t[1] / nSteps, acc_t[2] / nSteps, acc_t[3] / nSteps, acc_t[4] / nSteps);
    else if (prof && nSteps > 0)
        printf("loop profile (ms per step): inputs+predict %.3f, update %.3f, state extraction %.3f, resample %.3f, log writing %.3f\n",
               acc_t[0] / nSteps, acc_t[1] / nSteps, acc_t[2] / nSteps, acc_t[3] / nSteps, acc_t[4] / nSteps);
    finish_logs();
    if (log_err) die("state log");
    phd_destroy(f);
    return 0;
}
