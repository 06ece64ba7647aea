// phdslam — command-line driver with the reference executable's contract (src/main.cpp:1442-1500,
// run_synth :1075-1322): argv[1] = config file in the reference's cfg/config.cfg format,
// argv[2] = "synth".  Reads <data_directory>/measurements.txt and controls.txt (optionally
// measurement_times.txt / control_times.txt for asynchronous inputs, traj.txt with
// follow_trajectory), runs the Rao-Blackwellised GM-PHD-SLAM filter on the GPU with the state
// resident on the device, and writes state_estimate%05d.log (README:31-39 5-line format, or HEAD's
// 7-line writeLog with --log7) and loopTime.log.
//
//   phdslam <config.cfg> [synth] [--out DIR] [--seed S] [--capacity C] [--steps K] [--log7]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <string>
#include <vector>

#include "phdfilter_compat.h"

static void die(const char* where)
{
    fprintf(stderr, "%s failed: %s\n", where, phd_last_error());
    exit(EXIT_FAILURE);
}
#define CHK(call) do { if ((call) != PHD_OK) die(#call); } while (0)

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s <config.cfg> [synth] [--out DIR] [--seed S] [--capacity C] [--steps K] [--log7]\n", argv[0]);
        return 2;
    }
    std::string out_dir = ".";
    uint64_t seed = 1;
    int capacity = 0, max_steps = -1;
    bool log7 = false;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "synth")) continue;
        if (!strcmp(argv[i], "disparity")) { fprintf(stderr, "the disparity pipeline is out of scope\n"); return 2; }
        if (!strcmp(argv[i], "--log7")) log7 = true;
        else if (i + 1 < argc && !strcmp(argv[i], "--out")) out_dir = argv[++i];
        else if (i + 1 < argc && !strcmp(argv[i], "--seed")) seed = strtoull(argv[++i], nullptr, 10);
        else if (i + 1 < argc && !strcmp(argv[i], "--capacity")) capacity = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--steps")) max_steps = atoi(argv[++i]);
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
    phd_options opt = {};
    opt.n_particles = config.n_particles;
    opt.map_capacity = capacity > 0 ? capacity : 512;
    opt.max_measurements = max_m < PHD_MAX_MEASUREMENTS ? max_m : PHD_MAX_MEASUREMENTS;
    phd_filter* f = nullptr;
    CHK(phd_create(&config, &opt, &f));   // particles start at the configured pose, weights -log N (:1130-1145)
    const int kshot = config.nPredictParticles > 1 ? config.nPredictParticles : 1;
    const int n_max = config.n_particles * (kshot > 1 ? 5 * kshot : 1);

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
    for (int n = 0; n < nSteps; ++n) {
        timeval t0, t1;
        gettimeofday(&t0, nullptr);
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
                for (int i = 0; i < n_cur * kshot; ++i) {                                    // phdfilter.cu:1147-1152
                    noise[i].n_alpha = (float)(config.stdAlpha * randn());
                    noise[i].n_encoder = (float)(config.stdEncoder * randn());
                }
                CHK(phd_predict_ackerman(f, current_control, noise.data()));
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
        if (M > 0) CHK(phd_update(f, Z, M));                                                 // :1260-1272
        // state extraction (:1274) and log
        n_cur = phd_n_particles(f);
        phd_pose expected;
        int32_t n_map = 0, who = 0;
        CHK(phd_expected_pose(f, &expected));
        CHK(phd_map_estimate(f, map.data(), (int)map.size(), &n_map, &who));
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
        CHK(phd_get_particles(f, poses.data(), logw.data()));
        if (config.filterType == 1) {                                                        // cn_estimate (:360)
            cn_est.resize((size_t)phd_cardinality_length(f));
            CHK(phd_cardinality_estimate(f, cn_est.data(), nullptr));
        }
        // nEff test and resampling (:1281-1297)
        int32_t did = 0;
        CHK(phd_resample_if_needed(f, randu01(), M > 0, &did, ridx.data()));
        const bool cphd = config.filterType == 1;
        if (log7) {
            // weights and poses are those of the step (before resampling); a resample that shrinks a grown
            // (shotgun) particle set yields fewer parent indices than particles: the rest is marked -1
            if (did) for (int i = phd_n_particles(f); i < n_cur; ++i) ridx[i] = -1;
            if (cphd)
                CHK(phd_write_state_log7_cphd(out_dir.c_str(), n, &expected, map.data(), n_map, logw.data(), poses.data(),
                                              ridx.data(), n_cur, kshot, cn_est.data(), (int)cn_est.size()));
            else
                CHK(phd_write_state_log7(out_dir.c_str(), n, &expected, map.data(), n_map, logw.data(), poses.data(),
                                         ridx.data(), n_cur, config.maxCardinality, kshot));
        } else if (cphd) {
            CHK(phd_write_state_log_cphd(out_dir.c_str(), n, &expected, map.data(), n_map, logw.data(), poses.data(), n_cur,
                                         cn_est.data(), (int)cn_est.size()));
        } else {
            CHK(phd_write_state_log(out_dir.c_str(), n, &expected, map.data(), n_map, logw.data(), poses.data(), n_cur,
                                    config.maxCardinality));
        }
        uint32_t st = 0;
        if (phd_device_status(f, &st, nullptr, nullptr) != PHD_OK) die("capacity check");
        gettimeofday(&t1, nullptr);
        double elapsed = (t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_usec - t0.tv_usec) / 1000.0;
        if (FILE* tf = fopen(timefile.c_str(), "a")) { fprintf(tf, "%g\n", elapsed); fclose(tf); } // :1300-1305
        float ne = 0;
        if (phd_neff(f, &ne) != PHD_OK) { printf("nan weights detected! exiting...\n"); break; }  // :1307-1311
        printf("****** Time Step [%d/%d] ****** M=%d particles=%d map=%d resampled=%d %.3f ms\n", n, nSteps, M, n_cur, n_map, did, elapsed);
    }
    phd_destroy(f);
    return 0;
}
