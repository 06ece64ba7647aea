// phdslam — command-line driver with the reference executable's contract (src/main.cpp:1442-1500,
// run_synth :1075-1322): argv[1] = config file in the reference's cfg/config.cfg format,
// argv[2] = "synth".  Reads <data_directory>/measurements.txt and controls.txt, runs the
// Rao-Blackwellised GM-PHD-SLAM filter on the GPU with the state resident on the device, and
// writes state_estimate%05d.log (the 5-line format of README:31-39) and loopTime.log.
//
//   phdslam <config.cfg> [synth] [--out DIR] [--seed S] [--capacity C] [--steps K]
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
        fprintf(stderr, "usage: %s <config.cfg> [synth] [--out DIR] [--seed S] [--capacity C] [--steps K]\n", argv[0]);
        return 2;
    }
    std::string out_dir = ".";
    uint64_t seed = 1;
    int capacity = 0, max_steps = -1;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "synth")) continue;
        if (!strcmp(argv[i], "disparity")) { fprintf(stderr, "the disparity pipeline is out of scope\n"); return 2; }
        if (i + 1 < argc && !strcmp(argv[i], "--out")) out_dir = argv[++i];
        else if (i + 1 < argc && !strcmp(argv[i], "--seed")) seed = strtoull(argv[++i], nullptr, 10);
        else if (i + 1 < argc && !strcmp(argv[i], "--capacity")) capacity = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--steps")) max_steps = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    char data_dir[4096];
    int32_t n_steps_cfg = -1;
    CHK(phd_config_load(argv[1], &config, data_dir, sizeof(data_dir), &n_steps_cfg));       // main.cpp:1463
    phd_compat_seed_rng(seed);

    // inputs (main.cpp:1078-1088): data_directory must end with '/' (string concatenation, :1079)
    const std::string mfile = std::string(data_dir) + "measurements.txt", cfile = std::string(data_dir) + "controls.txt";
    size_t n_steps = 0, n_total = 0, n_ctrl = 0;
    CHK(phd_load_measurements(mfile.c_str(), 0, nullptr, 0, nullptr, 0, &n_steps, &n_total));
    std::vector<phd_measurement> meas(n_total ? n_total : 1);
    std::vector<int32_t> sizes(n_steps ? n_steps : 1);
    CHK(phd_load_measurements(mfile.c_str(), 0, meas.data(), meas.size(), sizes.data(), sizes.size(), &n_steps, &n_total));
    CHK(phd_load_controls(cfile.c_str(), -1, nullptr, 0, &n_ctrl));
    std::vector<phd_ackerman_control> controls(n_ctrl ? n_ctrl : 1);
    CHK(phd_load_controls(cfile.c_str(), -1, controls.data(), controls.size(), &n_ctrl));
    printf("Loaded %zu measurement steps, %zu control inputs\n", n_steps, n_ctrl);
    int nSteps = (int)n_steps;                                                               // :1096-1119
    if (n_steps_cfg > 0 && nSteps > n_steps_cfg) nSteps = n_steps_cfg;
    if (max_steps > 0 && nSteps > max_steps) nSteps = max_steps;

    int max_m = 1;
    for (int k = 0; k < nSteps; ++k) max_m = sizes[k] > max_m ? sizes[k] : max_m;
    phd_options opt = {};
    opt.n_particles = config.n_particles;
    opt.map_capacity = capacity > 0 ? capacity : 512;
    opt.max_measurements = max_m < PHD_MAX_MEASUREMENTS ? max_m : PHD_MAX_MEASUREMENTS;
    phd_filter* f = nullptr;
    CHK(phd_create(&config, &opt, &f));   // particles start at the configured pose, weights -log N (:1130-1145)
    const int N = config.n_particles;

    std::vector<phd_pose> poses(N);
    std::vector<float> logw(N);
    std::vector<phd_gaussian2d> map(opt.map_capacity);
    std::vector<phd_ackerman_noise> noise(N);
    std::string timefile = out_dir + "/loopTime.log";
    printf("STARTING SIMULATION\n");
    size_t moff = 0;
    for (int n = 0; n < nSteps; ++n) {
        timeval t0, t1;
        gettimeofday(&t0, nullptr);
        const int M = sizes[n];
        if (n > 0) {                                                                         // no motion at step 0 (:1244)
            if ((size_t)(n - 1) >= n_ctrl) { fprintf(stderr, "not enough controls\n"); break; }
            for (int s = 0; s < (config.subdividePredict > 0 ? config.subdividePredict : 1); ++s) {
                for (auto& nz : noise) {                                                     // phdfilter.cu:1147-1152
                    nz.n_alpha = (float)(config.stdAlpha * randn());
                    nz.n_encoder = (float)(config.stdEncoder * randn());
                }
                CHK(phd_predict_ackerman(f, controls[n - 1], noise.data()));                 // lock-step: U[n-1] (:1234)
            }
        }
        if (M > 0) CHK(phd_update(f, meas.data() + moff, M));                                // :1260-1272
        moff += M;
        // state extraction (:1274) and log (README:31-39)
        phd_pose expected;
        int32_t n_map = 0, who = 0;
        CHK(phd_expected_pose(f, &expected));
        CHK(phd_map_estimate(f, map.data(), (int)map.size(), &n_map, &who));
        CHK(phd_get_particles(f, poses.data(), logw.data()));
        CHK(phd_write_state_log(out_dir.c_str(), n, &expected, map.data(), n_map, logw.data(), poses.data(), N,
                                config.maxCardinality));
        // nEff test and resampling (:1281-1297)
        int32_t did = 0;
        CHK(phd_resample_if_needed(f, randu01(), M > 0, &did, nullptr));
        uint32_t st = 0;
        if (phd_device_status(f, &st, nullptr, nullptr) != PHD_OK) die("capacity check");
        gettimeofday(&t1, nullptr);
        double elapsed = (t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_usec - t0.tv_usec) / 1000.0;
        if (FILE* tf = fopen(timefile.c_str(), "a")) { fprintf(tf, "%g\n", elapsed); fclose(tf); } // :1300-1305
        float ne = 0;
        if (phd_neff(f, &ne) != PHD_OK) { printf("nan weights detected! exiting...\n"); break; }  // :1307-1311
        printf("****** Time Step [%d/%d] ****** M=%d map=%d resampled=%d %.3f ms\n", n, nSteps, M, n_map, did, elapsed);
    }
    phd_destroy(f);
    return 0;
}
