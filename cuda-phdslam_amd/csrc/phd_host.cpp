// phd_host.cpp — host-side boundary of the drop-in: cfg/config.cfg parser, text loaders,
// state_estimate%05d.log writer.  No device code, no Boost.
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <charconv>
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "phdslam.h"

extern "C" const char* phd_last_error(void);
// defined in phd_api.cpp
namespace { int host_fail(int code, const std::string& msg); }

// phd_api.cpp owns the thread-local error string; reach it through a tiny setter
extern "C" int phd_internal_set_error(int code, const char* msg);
namespace {
int host_fail(int code, const std::string& msg) { return phd_internal_set_error(code, msg.c_str()); }

enum KeyType { K_FLOAT, K_INT, K_BOOL, K_STRING, K_NSTEPS };
struct Key {
    KeyType type;
    size_t offset;  // into phd_slam_config
    double dflt;
};

#define OFF(field) offsetof(phd_slam_config, field)

// keys and defaults of loadConfig (src/main.cpp:960-1048).  The reference maps initial_vz onto
// vy0 and initial_vroll/initial_vpitch onto vyaw0 (:970-972); they go to their own fields here.
const std::map<std::string, Key>& key_table()
{
    static const std::map<std::string, Key> t = {
        {"debug", {K_BOOL, OFF(debug), 0}},
        {"initial_x", {K_FLOAT, OFF(x0), 0}},
        {"initial_y", {K_FLOAT, OFF(y0), 0}},
        {"initial_z", {K_FLOAT, OFF(z0), 0}},
        {"initial_roll", {K_FLOAT, OFF(roll0), 0}},
        {"initial_pitch", {K_FLOAT, OFF(pitch0), 0}},
        {"initial_yaw", {K_FLOAT, OFF(yaw0), 0}},
        {"initial_vx", {K_FLOAT, OFF(vx0), 0}},
        {"initial_vy", {K_FLOAT, OFF(vy0), 0}},
        {"initial_vz", {K_FLOAT, OFF(vz0), 0}},
        {"initial_vroll", {K_FLOAT, OFF(vroll0), 0}},
        {"initial_vpitch", {K_FLOAT, OFF(vpitch0), 0}},
        {"initial_vyaw", {K_FLOAT, OFF(vyaw0), 0}},
        {"follow_trajectory", {K_BOOL, OFF(followTrajectory), 0}},
        {"motion_type", {K_INT, OFF(motionType), 1}},
        {"acc_x", {K_FLOAT, OFF(ax), 0.5}},
        {"acc_y", {K_FLOAT, OFF(ay), 0}},
        {"acc_z", {K_FLOAT, OFF(az), 0}},
        {"acc_roll", {K_FLOAT, OFF(aroll), 0.0087}},
        {"acc_pitch", {K_FLOAT, OFF(apitch), 0.0087}},
        {"acc_yaw", {K_FLOAT, OFF(ayaw), 0.0087}},
        {"dt", {K_FLOAT, OFF(dt), 0.1}},
        {"max_bearing", {K_FLOAT, OFF(maxBearing), M_PI}},
        {"min_range", {K_FLOAT, OFF(minRange), 0}},
        {"max_range", {K_FLOAT, OFF(maxRange), 20}},
        {"std_bearing", {K_FLOAT, OFF(stdBearing), 0.0524}},
        {"std_range", {K_FLOAT, OFF(stdRange), 1.0}},
        {"clutter_rate", {K_FLOAT, OFF(clutterRate), 15}},
        {"pd", {K_FLOAT, OFF(pd), 0.98}},
        {"ps", {K_FLOAT, OFF(ps), 0.98}},
        {"n_particles", {K_INT, OFF(n_particles), 512}},
        {"n_predict_particles", {K_INT, OFF(nPredictParticles), 1}},
        {"resample_threshold", {K_FLOAT, OFF(resampleThresh), 0.15}},
        {"subdivide_predict", {K_INT, OFF(subdividePredict), 1}},
        {"birth_weight", {K_FLOAT, OFF(birthWeight), 0.05}},
        {"birth_noise_factor", {K_FLOAT, OFF(birthNoiseFactor), 1.5}},
        {"gate_births", {K_BOOL, OFF(gateBirths), 1}},
        {"gate_measurements", {K_BOOL, OFF(gateMeasurements), 1}},
        {"gate_threshold", {K_FLOAT, OFF(gateThreshold), 10}},
        {"feature_model", {K_INT, OFF(featureModel), 0}},
        {"min_expected_feature_weight", {K_FLOAT, OFF(minExpectedFeatureWeight), 0.33}},
        {"min_separation", {K_FLOAT, OFF(minSeparation), 5}},
        {"max_features", {K_INT, OFF(maxFeatures), 100}},
        {"min_feature_weight", {K_FLOAT, OFF(minFeatureWeight), 0.00001}},
        {"particle_weighting", {K_INT, OFF(particleWeighting), 1}},
        {"daughter_mixture_type", {K_INT, OFF(daughterMixtureType), 0}},
        {"n_samples", {K_INT, OFF(nSamples), 50}},
        {"max_cardinality", {K_INT, OFF(maxCardinality), 256}},
        {"filter_type", {K_INT, OFF(filterType), 1}},
        {"map_estimate", {K_INT, OFF(mapEstimate), 1}},
        {"cphd_disttype", {K_INT, OFF(cphdDistType), 0}},
        {"nu", {K_FLOAT, OFF(nu), 1}},
        {"distance_metric", {K_INT, OFF(distanceMetric), 0}},
        {"h", {K_FLOAT, OFF(h), 0}},
        {"l", {K_FLOAT, OFF(l), 0}},
        {"a", {K_FLOAT, OFF(a), 0}},
        {"b", {K_FLOAT, OFF(b), 0}},
        {"std_encoder", {K_FLOAT, OFF(stdEncoder), 0}},
        {"std_alpha", {K_FLOAT, OFF(stdAlpha), 0}},
        {"std_vx_features", {K_FLOAT, OFF(stdVxMap), 0}},
        {"std_vy_features", {K_FLOAT, OFF(stdVyMap), 0}},
        {"std_ax_features", {K_FLOAT, OFF(stdAxMap), 0}},
        {"std_ay_features", {K_FLOAT, OFF(stdAyMap), 0}},
        {"cov_vx_birth", {K_FLOAT, OFF(covVxBirth), 0}},
        {"cov_vy_birth", {K_FLOAT, OFF(covVyBirth), 0}},
        {"std_u", {K_FLOAT, OFF(stdU), 1}},
        {"std_v", {K_FLOAT, OFF(stdV), 1}},
        {"disparity_birth", {K_FLOAT, OFF(disparityBirth), 1000}},
        {"image_width", {K_INT, OFF(imageWidth), 600}},
        {"image_height", {K_INT, OFF(imageHeight), 480}},
        {"std_d_birth", {K_FLOAT, OFF(stdDBirth), 300}},
        {"fx", {K_FLOAT, OFF(fx), 1000}},
        {"fy", {K_FLOAT, OFF(fy), 1000}},
        {"u0", {K_FLOAT, OFF(u0), 512}},
        {"v0", {K_FLOAT, OFF(v0), 384}},
        {"particles_per_feature", {K_INT, OFF(particlesPerFeature), 100}},
        {"tau", {K_FLOAT, OFF(tau), 0}},
        {"beta", {K_FLOAT, OFF(beta), 1}},
        {"labeled_measurements", {K_BOOL, OFF(labeledMeasurements), 0}},
        {"data_directory", {K_STRING, 0, 0}},
        {"max_time_steps", {K_INT, OFF(maxSteps), 10000}},
        {"save_all_maps", {K_BOOL, OFF(saveAllMaps), 0}},
        {"save_prediction", {K_BOOL, OFF(savePrediction), 0}},
        {"n_steps", {K_NSTEPS, 0, -1}},
    };
    return t;
}

void store(phd_slam_config* cfg, const Key& k, double v)
{
    unsigned char* base = (unsigned char*)cfg;
    switch (k.type) {
    case K_FLOAT: *(float*)(base + k.offset) = (float)v; break;
    case K_INT: *(int32_t*)(base + k.offset) = (int32_t)v; break;
    case K_BOOL: *(uint8_t*)(base + k.offset) = v != 0 ? 1 : 0; break;
    default: break;
    }
}

std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

void derive(phd_slam_config* cfg)
{
    // src/main.cpp:1065-1066
    cfg->clutterDensity = cfg->clutterRate / (2 * cfg->maxBearing * cfg->maxRange);
}
} // namespace

extern "C" int phd_config_defaults(phd_slam_config* cfg)
{
    if (!cfg) return host_fail(PHD_ERR_INVALID_ARG, "null config");
    memset(cfg, 0, sizeof(*cfg));
    for (const auto& kv : key_table()) store(cfg, kv.second, kv.second.dflt);
    derive(cfg);
    return PHD_OK;
}

extern "C" int phd_config_load(const char* path, phd_slam_config* cfg, char* data_dir_out, size_t data_dir_cap,
                               int32_t* n_steps_out)
{
    if (!path || !cfg) return host_fail(PHD_ERR_INVALID_ARG, "null argument");
    std::ifstream ifs(path);
    if (!ifs) return host_fail(PHD_ERR_IO, std::string("Unable to open config file: ") + path);
    phd_config_defaults(cfg);
    std::string data_dir = "data/"; // src/main.cpp:1040
    int n_steps = -1;
    std::string line;
    int lineno = 0;
    while (std::getline(ifs, line)) {
        lineno++;
        size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        line = trim(line);
        if (line.empty()) continue;
        if (line[0] == '[') continue; // INI section headers carry no meaning for these keys
        size_t eq = line.find('=');
        if (eq == std::string::npos)
            return host_fail(PHD_ERR_PARSE, std::string(path) + ":" + std::to_string(lineno) + ": expected key = value");
        std::string key = trim(line.substr(0, eq));
        std::string val = trim(line.substr(eq + 1));
        auto it = key_table().find(key);
        if (it == key_table().end())
            return host_fail(PHD_ERR_PARSE, std::string(path) + ":" + std::to_string(lineno) + ": unknown key '" + key + "'");
        const Key& k = it->second;
        if (k.type == K_STRING) { data_dir = val; continue; }
        double v = 0;
        if (k.type == K_BOOL && (val == "true" || val == "on" || val == "yes")) v = 1;
        else if (k.type == K_BOOL && (val == "false" || val == "off" || val == "no")) v = 0;
        else {
            char* end = nullptr;
            errno = 0;
            v = strtod(val.c_str(), &end);
            if (end == val.c_str() || *end != '\0' || errno == ERANGE)
                return host_fail(PHD_ERR_PARSE, std::string(path) + ":" + std::to_string(lineno) + ": bad value '" + val +
                                                    "' for key '" + key + "'");
        }
        if (k.type == K_NSTEPS) n_steps = (int)v;
        else store(cfg, k, v);
    }
    derive(cfg);
    if (data_dir_out && data_dir_cap) {
        if (data_dir.size() + 1 > data_dir_cap) return host_fail(PHD_ERR_CAPACITY, "data_directory too long");
        memcpy(data_dir_out, data_dir.c_str(), data_dir.size() + 1);
    }
    if (n_steps_out) *n_steps_out = n_steps;
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// loaders
// ---------------------------------------------------------------------------------------------
namespace {
// split a line into numbers; ',' ';' and whitespace all separate (bundled python/controls_synth.txt
// is comma separated)
bool numbers_of(const std::string& line, std::vector<double>& out)
{
    out.clear();
    const char* p = line.c_str();
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == ',' || *p == ';' || *p == '\r') p++;
        if (!*p) break;
        char* end = nullptr;
        double v = strtod(p, &end);
        if (end == p) return false;
        out.push_back(v);
        p = end;
    }
    return true;
}
bool is_header(const std::string& line)
{
    std::string t = trim(line);
    if (t.empty()) return false;
    return !(isdigit((unsigned char)t[0]) || t[0] == '-' || t[0] == '+' || t[0] == '.');
}
} // namespace

extern "C" int phd_load_measurements(const char* path, int triples, phd_measurement* out, size_t out_capacity,
                                     int32_t* step_sizes_out, size_t steps_capacity, size_t* n_steps_out,
                                     size_t* n_total_out)
{
    if (!path) return host_fail(PHD_ERR_INVALID_ARG, "null path");
    std::ifstream f(path);
    if (!f) return host_fail(PHD_ERR_IO, std::string("could not open measurements file: ") + path);
    std::string line;
    std::vector<double> nums;
    size_t n_steps = 0, n_total = 0;
    bool first = true;
    const int stride = triples ? 3 : 2;
    std::vector<std::pair<size_t, std::string>> pending_blank; // blank lines inside the file are empty steps
    size_t blanks = 0;
    while (std::getline(f, line)) {
        if (first) {
            first = false;
            // the reference always skips line 1 as a header (src/main.cpp:227); the bundled
            // python/measurements_synth.txt has none: skip only a non-numeric line
            if (is_header(line)) continue;
        }
        if (trim(line).empty()) { blanks++; continue; } // trailing blank lines are dropped (:238)
        // blank lines before a data line were real (empty) steps
        for (; blanks > 0; --blanks) {
            if (step_sizes_out && n_steps < steps_capacity) step_sizes_out[n_steps] = 0;
            n_steps++;
        }
        if (!numbers_of(line, nums) || nums.size() % stride != 0)
            return host_fail(PHD_ERR_PARSE, std::string(path) + ": step " + std::to_string(n_steps) + ": expected " +
                                                (triples ? "range bearing label triples" : "range bearing pairs"));
        const size_t m = nums.size() / stride;
        for (size_t i = 0; i < m; ++i) {
            if (out && n_total + i < out_capacity) {
                out[n_total + i].range = (float)nums[i * stride];
                out[n_total + i].bearing = (float)nums[i * stride + 1];
                out[n_total + i].label = triples ? (int32_t)nums[i * stride + 2] : 0;
            }
        }
        if (step_sizes_out && n_steps < steps_capacity) step_sizes_out[n_steps] = (int32_t)m;
        n_steps++;
        n_total += m;
    }
    if (n_steps_out) *n_steps_out = n_steps;
    if (n_total_out) *n_total_out = n_total;
    if (out && n_total > out_capacity) return host_fail(PHD_ERR_CAPACITY, "measurement buffer too small");
    if (step_sizes_out && n_steps > steps_capacity) return host_fail(PHD_ERR_CAPACITY, "step-size buffer too small");
    return PHD_OK;
}

extern "C" int phd_load_controls(const char* path, int has_header, phd_ackerman_control* out, size_t capacity,
                                 size_t* n_out)
{
    if (!path) return host_fail(PHD_ERR_INVALID_ARG, "null path");
    std::ifstream f(path);
    if (!f) return host_fail(PHD_ERR_IO, std::string("could not open controls file: ") + path);
    std::string line;
    std::vector<double> nums;
    size_t n = 0;
    bool first = true;
    while (std::getline(f, line)) {
        if (first) {
            first = false;
            if (has_header > 0 || (has_header < 0 && is_header(line))) continue; // src/main.cpp:177
        }
        if (trim(line).empty()) continue; // the reference pushes a garbage final entry here (:178-186)
        if (!numbers_of(line, nums) || nums.size() < 2)
            return host_fail(PHD_ERR_PARSE, std::string(path) + ": control " + std::to_string(n) + ": expected 'v_encoder alpha'");
        if (out && n < capacity) { out[n].v_encoder = (float)nums[0]; out[n].alpha = (float)nums[1]; } // :182
        n++;
    }
    if (n_out) *n_out = n;
    if (out && n > capacity) return host_fail(PHD_ERR_CAPACITY, "control buffer too small");
    return PHD_OK;
}

// loadTimestamps (src/main.cpp:147-166): one value per line, NO header line, trailing blank dropped
extern "C" int phd_load_timestamps(const char* path, float* out, size_t capacity, size_t* n_out)
{
    if (!path) return host_fail(PHD_ERR_INVALID_ARG, "null path");
    std::ifstream f(path);
    size_t n = 0;
    if (f) { // a missing file means "no timestamps" (lock-step mode), like the reference
        std::string line;
        std::vector<double> nums;
        while (std::getline(f, line)) {
            if (trim(line).empty()) continue;
            if (!numbers_of(line, nums) || nums.empty())
                return host_fail(PHD_ERR_PARSE, std::string(path) + ": line " + std::to_string(n + 1) + ": expected a time stamp");
            if (out && n < capacity) out[n] = (float)nums[0];
            n++;
        }
    }
    if (n_out) *n_out = n;
    if (out && n > capacity) return host_fail(PHD_ERR_CAPACITY, "timestamp buffer too small");
    return PHD_OK;
}

// loadTrajectory (src/main.cpp:242-260): "px py ptheta vx vy vtheta" per line, '%' lines skipped
extern "C" int phd_load_trajectory(const char* path, phd_pose* out, size_t capacity, size_t* n_out)
{
    if (!path) return host_fail(PHD_ERR_INVALID_ARG, "null path");
    std::ifstream f(path);
    if (!f) return host_fail(PHD_ERR_IO, std::string("could not open trajectory file: ") + path);
    std::string line;
    std::vector<double> nums;
    size_t n = 0;
    while (std::getline(f, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '%') continue;
        if (!numbers_of(line, nums) || nums.size() < 3)
            return host_fail(PHD_ERR_PARSE, std::string(path) + ": pose " + std::to_string(n) + ": expected 'px py ptheta [vx vy vtheta]'");
        if (out && n < capacity) {
            phd_pose q = {(float)nums[0], (float)nums[1], (float)nums[2], 0, 0, 0};
            if (nums.size() >= 6) { q.vx = (float)nums[3]; q.vy = (float)nums[4]; q.vtheta = (float)nums[5]; }
            out[n] = q;
        }
        n++;
    }
    if (n_out) *n_out = n;
    if (out && n > capacity) return host_fail(PHD_ERR_CAPACITY, "trajectory buffer too small");
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// Number formatting of the state logs.  The reference streams every float through operator<< (src/main.cpp:861-952): the
// default ostream format, i.e. printf's %g with precision 6.  std::to_chars(general, 6) produces exactly those characters
// (C++17 [charconv.to.chars]: "as if by std::printf" with the same conversion) without the locale / sentry / virtual-call
// machinery of a stream: at 4096 particles a log is ~29 000 numbers, which cost the writer threads 3 - 6 ms through a
// stream — more than six filter steps — and made the log queue's back-pressure the p90 of the driver's loop time.
// ---------------------------------------------------------------------------------------------
struct LogBuf {
    std::string s;
    void reserve(size_t n) { s.reserve(n); }
    void num(float v)
    {
        char tmp[32];
        const auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::general, 6);
        s.append(tmp, (size_t)(r.ptr - tmp));
        s.push_back(' ');
    }
    void integer(int v)
    {
        char tmp[16];
        const auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
        s.append(tmp, (size_t)(r.ptr - tmp));
        s.push_back(' ');
    }
    void endl() { s.push_back('\n'); }
};

static std::string state_log_name(const char* dir, int step)
{
    std::ostringstream name;
    if (dir && dir[0]) {
        name << dir;
        if (name.str().back() != '/') name << '/';
    }
    name << "state_estimate" << std::setfill('0') << std::setw(5) << step << ".log"; // src/main.cpp:854-858
    return name.str();
}

static int write_whole(const std::string& name, const std::string& text, bool append)
{
    FILE* f = fopen(name.c_str(), append ? "ab" : "wb");
    if (!f) return host_fail(PHD_ERR_IO, "cannot write " + name);
    const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
    const bool closed = fclose(f) == 0;
    return (ok && closed) ? PHD_OK : host_fail(PHD_ERR_IO, "write failed: " + name);
}

// ---------------------------------------------------------------------------------------------
// state_estimate%05d.log — the 5-line consumer contract (README:31-39; python/batch_analyze.py:16-24;
// python/plot_phdslam.py:205-226): default operator<< float formatting, space separated, trailing space
// ---------------------------------------------------------------------------------------------
static int write_log5(const char* dir, int step, const phd_pose* e, const phd_gaussian2d* map, int n_map,
                      const float* log_weights, const phd_pose* poses, int n_particles, int max_cardinality, const float* cn)
{
    if (!e) return host_fail(PHD_ERR_INVALID_ARG, "null pose");
    LogBuf o;
    o.reserve(64 + (size_t)n_map * 80 + (size_t)n_particles * 64 + (size_t)(max_cardinality + 1) * 12);
    o.num(e->px); o.num(e->py); o.num(e->ptheta); o.num(e->vx); o.num(e->vy); o.num(e->vtheta); o.endl();
    for (int n = 0; n < n_map; ++n) { // weight mx my c0 c1 c2 c3 (:866-878)
        o.num(map[n].weight);
        for (int i = 0; i < 2; ++i) o.num(map[n].mean[i]);
        for (int i = 0; i < 4; ++i) o.num(map[n].cov[i]);
    }
    o.endl();
    for (int n = 0; n < n_particles; ++n) o.num(log_weights[n]);
    o.endl();
    for (int n = 0; n < n_particles; ++n) {
        o.num(poses[n].px); o.num(poses[n].py); o.num(poses[n].ptheta); o.num(poses[n].vx); o.num(poses[n].vy); o.num(poses[n].vtheta);
    }
    o.endl();
    for (int n = 0; n < max_cardinality + 1; ++n) { // PHD: zeros, CPHD: cn_estimate (:942-949)
        if (cn) o.num(cn[n]);
        else o.s.append("0 ");
    }
    o.endl();
    return write_whole(state_log_name(dir, step), o.s, false);
}

extern "C" int phd_write_state_log(const char* dir, int step, const phd_pose* e, const phd_gaussian2d* map, int n_map,
                                   const float* log_weights, const phd_pose* poses, int n_particles, int max_cardinality)
{
    return write_log5(dir, step, e, map, n_map, log_weights, poses, n_particles, max_cardinality, nullptr);
}

// CPHD (filter_type = 1): the last line carries cn_estimate[0..max_cardinality] (src/main.cpp:944-949)
extern "C" int phd_write_state_log_cphd(const char* dir, int step, const phd_pose* e, const phd_gaussian2d* map, int n_map,
                                        const float* log_weights, const phd_pose* poses, int n_particles,
                                        const float* cn_estimate, int cn_len)
{
    if (!cn_estimate || cn_len < 1) return host_fail(PHD_ERR_INVALID_ARG, "null cardinality estimate");
    return write_log5(dir, step, e, map, n_map, log_weights, poses, n_particles, cn_len - 1, cn_estimate);
}

// HEAD's writeLog (src/main.cpp:848-954): 7 lines — pose / static map / dynamic map (empty: static
// model) / log-weights / poses / resample indices / cardinality — opened in APPEND mode; at t = 0 the
// weights and poses are repeated nPredictParticles times (:901-934).
static int write_log7(const char* dir, int step, const phd_pose* e, const phd_gaussian2d* map, int n_map,
                      const float* log_weights, const phd_pose* poses, const int32_t* resample_idx,
                      int n_particles, int max_cardinality, int n_predict_particles, const float* cn)
{
    if (!e) return host_fail(PHD_ERR_INVALID_ARG, "null pose");
    LogBuf o;
    const int times = (step == 0 && n_predict_particles > 1) ? n_predict_particles : 1;
    o.reserve(64 + (size_t)n_map * 80 + (size_t)n_particles * (64 * (size_t)times + 8) + (size_t)(max_cardinality + 1) * 12);
    o.num(e->px); o.num(e->py); o.num(e->ptheta); o.num(e->vx); o.num(e->vy); o.num(e->vtheta); o.endl();
    for (int n = 0; n < n_map; ++n) {
        o.num(map[n].weight);
        for (int i = 0; i < 2; ++i) o.num(map[n].mean[i]);
        for (int i = 0; i < 4; ++i) o.num(map[n].cov[i]);
    }
    o.endl();
    o.endl(); // dynamic map: none in the static feature model
    for (int k = 0; k < times; ++k)
        for (int n = 0; n < n_particles; ++n) o.num(log_weights[n]);
    o.endl();
    for (int k = 0; k < times; ++k)
        for (int n = 0; n < n_particles; ++n) {
            o.num(poses[n].px); o.num(poses[n].py); o.num(poses[n].ptheta); o.num(poses[n].vx); o.num(poses[n].vy); o.num(poses[n].vtheta);
        }
    o.endl();
    for (int n = 0; n < n_particles; ++n) o.integer(resample_idx ? resample_idx[n] : n);
    o.endl();
    for (int n = 0; n < max_cardinality + 1; ++n) {
        if (cn) o.num(cn[n]);
        else o.s.append("0 ");
    }
    o.endl();
    return write_whole(state_log_name(dir, step), o.s, true);       // HEAD opens the log in append mode (:861)
}

extern "C" int phd_write_state_log7(const char* dir, int step, const phd_pose* e, const phd_gaussian2d* map, int n_map,
                                    const float* log_weights, const phd_pose* poses, const int32_t* resample_idx,
                                    int n_particles, int max_cardinality, int n_predict_particles)
{
    return write_log7(dir, step, e, map, n_map, log_weights, poses, resample_idx, n_particles, max_cardinality,
                      n_predict_particles, nullptr);
}

extern "C" int phd_write_state_log7_cphd(const char* dir, int step, const phd_pose* e, const phd_gaussian2d* map, int n_map,
                                         const float* log_weights, const phd_pose* poses, const int32_t* resample_idx,
                                         int n_particles, int n_predict_particles, const float* cn_estimate, int cn_len)
{
    if (!cn_estimate || cn_len < 1) return host_fail(PHD_ERR_INVALID_ARG, "null cardinality estimate");
    return write_log7(dir, step, e, map, n_map, log_weights, poses, resample_idx, n_particles, cn_len - 1,
                      n_predict_particles, cn_estimate);
}
