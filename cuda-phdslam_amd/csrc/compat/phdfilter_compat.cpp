// phdfilter_compat.cpp — src/phdfilter.h's entry points implemented over the C-ABI.
//
// Semantics follow the reference: the caller owns the SynthSLAM, every call uploads the state it
// needs, runs the gfx950 kernels and writes the results back into the SynthSLAM (the reference
// round-trips the whole state per call as well, src/phdfilter.cu:2952-3002,3293-3318).  For
// throughput, drive the C-ABI directly and keep the state on the device (see phdslam_main.cpp).
#include "phdfilter_compat.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

static_assert(sizeof(ConstantVelocityState) == 24 && sizeof(Gaussian2D) == 28 && sizeof(RangeBearingMeasurement) == 12 &&
                  sizeof(SlamConfig) == 324 && sizeof(AckermanControl) == 8,
              "reference layouts");

SlamConfig config;

namespace {
phd_filter* g_filter = nullptr;
int g_n = 0, g_cap = 0;

[[noreturn]] void die(const char* where)
{
    // the reference's error convention: checkCudaErrors -> print + exit(EXIT_FAILURE)
    fprintf(stderr, "%s failed: %s\n", where, phd_last_error());
    exit(EXIT_FAILURE);
}
#define CHK(call) do { if ((call) != PHD_OK) die(#call); } while (0)

int needed_capacity(const SynthSLAM& p, int n_meas)
{
    size_t mx = 0;
    for (const auto& m : p.maps_static) mx = std::max(mx, m.size());
    // births can add one component per measurement and step; leave head room
    int want = (int)mx + 2 * std::max(n_meas, 32) + 64;
    int cap = 256;
    while (cap < want) cap *= 2;
    return cap;
}

// The filter is created for the CONFIGURED particle count; with the particle shotgun (n_predict_particles = k > 1,
// src/phdfilter.cu:1185-1238) the caller's SynthSLAM grows by k per phdPredict and shrinks back in
// resampleParticles(particles, config.n_particles) (src/main.cpp:1289): the library holds room for
// 5 n_particles k particles, and the count it works on follows the SynthSLAM of each call.
void ensure_filter(const SynthSLAM& p, int n_meas)
{
    const int cap = needed_capacity(p, n_meas);
    const int k = config.nPredictParticles > 1 ? config.nPredictParticles : 1;
    int base = p.n_particles;
    if (k > 1 && config.n_particles > 0 && p.n_particles >= config.n_particles && p.n_particles <= 5 * config.n_particles * k)
        base = config.n_particles;
    if (!(g_filter && g_n == base && cap <= g_cap)) {
        if (g_filter) phd_destroy(g_filter);
        g_filter = nullptr;
        phd_options o = {};
        o.n_particles = base;
        o.map_capacity = std::max(cap, g_cap);
        CHK(phd_create(&config, &o, &g_filter));
        g_n = base;
        g_cap = o.map_capacity;
    }
    if (phd_n_particles(g_filter) != p.n_particles) CHK(phd_set_particle_count(g_filter, p.n_particles));
}

void upload(const SynthSLAM& p)
{
    CHK(phd_set_particles(g_filter, p.states.data(), p.weights.data(), p.n_particles));
    std::vector<Gaussian2D> concat;
    std::vector<int32_t> sizes(p.n_particles);
    for (int i = 0; i < p.n_particles; ++i) {
        sizes[i] = (int32_t)p.maps_static[i].size();
        concat.insert(concat.end(), p.maps_static[i].begin(), p.maps_static[i].end());
    }
    CHK(phd_set_maps(g_filter, concat.data(), sizes.data()));
    if (config.filterType == 1) {
        // CPHD: SynthSLAM::cardinalities (src/slamtypes.h:296), rows of maxCardinality+1 log-probabilities;
        // rows the caller never sized start uniform (src/main.cpp:1140-1143)
        const int k = phd_cardinality_length(g_filter);
        std::vector<float> cn((size_t)p.n_particles * k, -logf((float)k));
        for (int i = 0; i < p.n_particles; ++i)
            if ((int)p.cardinalities[i].size() == k) std::copy(p.cardinalities[i].begin(), p.cardinalities[i].end(), cn.begin() + (size_t)i * k);
        CHK(phd_set_cardinalities(g_filter, cn.data()));
    }
}

void download(SynthSLAM& p, bool maps)
{
    CHK(phd_get_particles(g_filter, p.states.data(), p.weights.data()));
    if (!maps) return;
    std::vector<int32_t> sizes(p.n_particles);
    CHK(phd_get_map_sizes(g_filter, sizes.data()));
    size_t total = 0;
    for (int s : sizes) total += s;
    std::vector<Gaussian2D> concat(total ? total : 1);
    CHK(phd_get_maps(g_filter, concat.data(), concat.size(), sizes.data()));
    size_t off = 0;
    for (int i = 0; i < p.n_particles; ++i) {
        p.maps_static[i].assign(concat.begin() + off, concat.begin() + off + sizes[i]);
        off += sizes[i];
    }
    if (config.filterType == 1) {
        const int k = phd_cardinality_length(g_filter);
        std::vector<float> cn((size_t)p.n_particles * k);
        CHK(phd_get_cardinalities(g_filter, cn.data()));
        for (int i = 0; i < p.n_particles; ++i) p.cardinalities[i].assign(cn.begin() + (size_t)i * k, cn.begin() + (size_t)(i + 1) * k);
    }
}
} // namespace

void initRandomNumberGenerators() {} // the variance diagnostic's cuRAND states: out of scope

void setDeviceConfig(const SlamConfig& c)
{
    config = c;
    if (g_filter) CHK(phd_set_config(g_filter, &config));
}

void phdPredict(SynthSLAM& particles, ...)
{
    if (config.motionType != ACKERMAN_MOTION) {
        fprintf(stderr, "phdPredict: only the Ackerman motion model is supported\n");
        exit(EXIT_FAILURE);
    }
    va_list ap;
    va_start(ap, particles);
    AckermanControl control = va_arg(ap, AckermanControl); // src/phdfilter.cu:1140-1144
    va_end(ap);
    ensure_filter(particles, 0);
    CHK(phd_set_particles(g_filter, particles.states.data(), particles.weights.data(), particles.n_particles));
    // host-drawn noise in the reference's order (n_alpha, n_encoder) per PREDICTED particle (src/phdfilter.cu:1147-1152)
    const int k = config.nPredictParticles > 1 ? config.nPredictParticles : 1;
    const int n = particles.n_particles, n_pred = n * k;
    std::vector<AckermanNoise> noise((size_t)n_pred);
    for (auto& nz : noise) {
        nz.n_alpha = (REAL)(config.stdAlpha * randn());
        nz.n_encoder = (REAL)(config.stdEncoder * randn());
    }
    CHK(phd_predict_ackerman(g_filter, control, noise.data()));
    if (k > 1) {
        // the reference grows the SynthSLAM (src/phdfilter.cu:1182-1234): every map, cardinality row and resample index
        // k times in place, weights - log k, n_particles = nPredict.  (The device shares the maps through its parent
        // indirection; the host copies are what the caller's SynthSLAM owns.)
        if (phd_n_particles(g_filter) != n_pred) die("phdPredict (particle shotgun)");
        std::vector<std::vector<Gaussian2D> > maps((size_t)n_pred);
        std::vector<std::vector<REAL> > card((size_t)n_pred);
        std::vector<int> ridx((size_t)n_pred);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < k; ++j) {
                maps[(size_t)i * k + j] = particles.maps_static[i];
                card[(size_t)i * k + j] = particles.cardinalities[i];
                ridx[(size_t)i * k + j] = particles.resample_idx[i];
            }
        particles.maps_static.swap(maps);
        particles.cardinalities.swap(card);
        particles.resample_idx.swap(ridx);
        particles.states.resize((size_t)n_pred);
        particles.weights.resize((size_t)n_pred);
        particles.variances.resize((size_t)n_pred);
        particles.n_particles = n_pred;
        CHK(phd_get_particles(g_filter, particles.states.data(), particles.weights.data()));   // weights - log k (:1213)
        return;
    }
    CHK(phd_get_particles(g_filter, particles.states.data(), nullptr));
}

SynthSLAM phdUpdateSynth(SynthSLAM& particles, measurementSet measurements)
{
    SynthSLAM particlesPreMerge(particles); // the reference returns a pre-update copy (src/phdfilter.cu:3351,3760)
    if (measurements.empty()) return particlesPreMerge;
    ensure_filter(particles, (int)measurements.size());
    upload(particles);
    CHK(phd_update(g_filter, measurements.data(), (int)measurements.size()));
    uint32_t st = 0;
    if (phd_device_status(g_filter, &st, nullptr, nullptr) != PHD_OK) die("phd_update (capacity)");
    download(particles, true);
    return particlesPreMerge;
}

REAL computeNeff(SynthSLAM& particles)
{
    ensure_filter(particles, 0);
    CHK(phd_set_particles(g_filter, nullptr, particles.weights.data(), particles.n_particles));
    REAL ne = 0;
    CHK(phd_neff(g_filter, &ne));
    return ne;
}

SynthSLAM resampleParticles(SynthSLAM oldParticles, int n_new_particles)
{
    // n_new == n, or a grown (shotgun) set back to the configured count (src/main.cpp:1289)
    const int n_old = oldParticles.n_particles;
    if (n_new_particles < 0) n_new_particles = n_old;
    const bool shrink = n_new_particles != n_old;
    if (shrink && !(config.nPredictParticles > 1 && n_new_particles == config.n_particles && n_old > n_new_particles)) {
        fprintf(stderr, "resampleParticles: the particle count can only change back to n_particles after a particle shotgun\n");
        exit(EXIT_FAILURE);
    }
    ensure_filter(oldParticles, 0);
    CHK(phd_set_particles(g_filter, nullptr, oldParticles.weights.data(), n_old));
    const double u = randu01();
    std::vector<int32_t> idx((size_t)n_old);
    CHK(phd_resample(g_filter, &u, 1, idx.data()));           // a grown set comes back with n_particles indices
    if (phd_n_particles(g_filter) != n_new_particles) die("resampleParticles (particle count)");
    return oldParticles.copy_particles(std::vector<int>(idx.begin(), idx.begin() + n_new_particles));
}

void recoverSlamState(SynthSLAM& particles, ConstantVelocityState& expectedPose, std::vector<REAL>& cn_estimate)
{
    ensure_filter(particles, 0);
    CHK(phd_set_particles(g_filter, particles.states.data(), particles.weights.data(), particles.n_particles));
    CHK(phd_expected_pose(g_filter, &expectedPose));
    // MAP map: the map of the arg-max-weight particle (src/main.cpp:344-361; map_estimate = 0 yields no
    // map for N > 1 in the reference — treated as MAP here)
    float best = -3.4e38f;
    int bi = 0;
    for (int i = 0; i < particles.n_particles; ++i)
        if (particles.weights[i] > best) { best = particles.weights[i]; bi = i; }
    particles.max_map_static = particles.maps_static[bi];
    cn_estimate = particles.cardinalities[bi];
    if ((config.mapEstimate & 2) && particles.n_particles > 1) {
        // expected a priori map estimate (src/main.cpp:363-379): computeExpectedMap on the device
        ensure_filter(particles, 0);
        upload(particles);
        std::vector<Gaussian2D> out(1024);
        int32_t n = 0;
        int rc = phd_expected_map(g_filter, out.data(), (int)out.size(), &n);
        if (rc == PHD_ERR_CAPACITY && n > (int)out.size()) {
            out.resize((size_t)n);
            rc = phd_expected_map(g_filter, out.data(), (int)out.size(), &n);
        }
        if (rc != PHD_OK) die("phd_expected_map");
        out.resize((size_t)n);
        particles.exp_map_static = out;
    }
}

// ---------------------------------------------------------------------------------------------
// rng.h: seedable portable generator (xoshiro256** seeded by splitmix64; Box-Muller normals)
// ---------------------------------------------------------------------------------------------
namespace {
uint64_t rs[4] = {0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0x2545F4914F6CDD1Dull};
bool have_spare = false;
double spare = 0;
inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t next_u64()
{
    const uint64_t result = rotl(rs[1] * 5, 7) * 9;
    const uint64_t t = rs[1] << 17;
    rs[2] ^= rs[0]; rs[3] ^= rs[1]; rs[1] ^= rs[2]; rs[0] ^= rs[3];
    rs[2] ^= t;
    rs[3] = rotl(rs[3], 45);
    return result;
}
} // namespace

extern "C" void phd_compat_seed_rng(uint64_t seed)
{
    for (int i = 0; i < 4; ++i) {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        rs[i] = z ^ (z >> 31);
    }
    have_spare = false;
}

extern "C" double randu01() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

extern "C" double randn()
{
    if (have_spare) { have_spare = false; return spare; }
    double u1;
    do { u1 = randu01(); } while (u1 <= 0.0);
    const double u2 = randu01();
    const double r = std::sqrt(-2.0 * std::log(u1));
    spare = r * std::sin(6.283185307179586 * u2);
    have_spare = true;
    return r * std::cos(6.283185307179586 * u2);
}
