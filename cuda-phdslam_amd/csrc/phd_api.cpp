// phd_api.cpp — the C-ABI of include/phdslam.h over the gfx950 kernels.
//
// One phd_filter owns every device buffer of one rank's particle shard for the whole run
// (the reference allocates and frees ~22 buffers per update, src/phdfilter.cu:2966-3102,
// 3403-3438,3525-3552).  All work is enqueued on one HIP stream; nothing in the per-step
// path allocates, frees or synchronises.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "phd_device.h"
#include "phdslam.h"

using namespace phd;

// layout checks against the reference's src/slamtypes.h (offsets measured from the reference header)
static_assert(sizeof(phd_gaussian2d) == 28, "Gaussian2D layout");
static_assert(sizeof(phd_pose) == 24, "ConstantVelocityState layout");
static_assert(sizeof(phd_measurement) == 12, "RangeBearingMeasurement layout");
static_assert(sizeof(phd_slam_config) == 324, "SlamConfig layout");
static_assert(offsetof(phd_slam_config, dt) == 80, "SlamConfig.dt");
static_assert(offsetof(phd_slam_config, minRange) == 84, "SlamConfig.minRange");
static_assert(offsetof(phd_slam_config, n_particles) == 196, "SlamConfig.n_particles");
static_assert(offsetof(phd_slam_config, birthWeight) == 212, "SlamConfig.birthWeight");
static_assert(offsetof(phd_slam_config, minSeparation) == 232, "SlamConfig.minSeparation");
static_assert(offsetof(phd_slam_config, filterType) == 260, "SlamConfig.filterType");
static_assert(offsetof(phd_slam_config, labeledMeasurements) == 292, "SlamConfig.labeledMeasurements");
static_assert(offsetof(phd_slam_config, l) == 296, "SlamConfig.l");
static_assert(offsetof(phd_slam_config, saveAllMaps) == 320, "SlamConfig.saveAllMaps");

static thread_local std::string g_err;

extern "C" const char* phd_last_error(void) { return g_err.c_str(); }

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

// used by phd_host.cpp (config / loaders / log writer) to report through the same channel
extern "C" int phd_internal_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }

#define HIPCHK(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(PHD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

struct TimedEvent {
    hipEvent_t a, b;
    int kind;
};

struct phd_filter {
    phd_slam_config cfg;
    DevConfig dcfg;
    int n = 0; // CURRENT particle count (n_base normally; grows by n_predict_particles per predict)
    int n_base = 0, n_max = 0;
    float* logw_alt = nullptr; // second log-weight buffer (shotgun predict writes out of place)
    int cap = 0, MM = 0, S_cap = 0, device = 0;
    int M_limit = 0;               // the caller's max_measurements: what a scan is clamped to (MM may be larger: rounded up to a compiled-in layout)
    // spill path: survivor lists longer than the LDS capacity (created with survivor_capacity > 2048): records in HBM and the
    // plain global-memory merge of phd_spill.h for the particles that need it
    int spill_cap = 0;
    float* spill_rec = nullptr;
    int* spill_meta = nullptr;
    unsigned short* spill_out = nullptr;
    long long* spill_acc = nullptr;
    int* spill_tmp = nullptr;
    int n_global = 0, global_offset = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t lds_bytes = 0;
    bool three_per_cu = false;     // the build of the update kernel this filter runs (80 registers, three workgroups per CU): decided at create
    int last_update_fn = -1;       // the instantiation the last update launch ran (phd_debug_update_instantiation)
    bool any_layout = false;       // PHD_LAYOUT=0 at create: never the instantiations with this filter's LDS layout compiled in (A/B, tests)

    // the map buffers are ONE allocation — [buffer 0 | buffer 1 | guests], n_max slabs each; the guests only on a shard — so
    // that the indirection, which counts slabs from the start of the CURRENT buffer, can name a slab of either buffer or a
    // guest: the copy-free resample of a shard (phd_global_resample_pull / _end, phd_kernels.hip) parks a remote parent in a
    // guest slab and leaves local ones where they are.  counts / CPHD rows: the same layout, so the same index serves all three.
    float* maps_arena = nullptr;
    int* counts_arena = nullptr;
    float* cn_arena = nullptr;
    float* maps_g = nullptr;      // guests (NULL: not a shard among several)
    int* counts_g = nullptr;
    float* cn_g = nullptr;
    bool copy_free = true;        // PHD_COPY_FREE=0: the copying forms (A/B, tests)
    int copy_free_resamples = 0;  // global resamples that took the copy-free form (diagnostics)
    float* maps[2] = {nullptr, nullptr};
    int* counts[2] = {nullptr, nullptr};
    int cur = 0;
    int* parent[3] = {nullptr, nullptr, nullptr}; // [pcur] live, [pcur^1] next, [2] frozen scratch
    int pcur = 0;
    bool parent_dirty = false;
    phd_pose* pose[3] = {nullptr, nullptr, nullptr}; // [pose_cur] live, other: next / scratch
    int pose_cur = 0;
    float* logw = nullptr;
    float* logw_scratch = nullptr;
    float* logw_raw = nullptr;      // where a local step leaves the un-normalised weights: logw_raw_own, or the caller's buffer (phd_set_raw_target)
    float* logw_raw_own = nullptr;
    float* dlogw = nullptr;
    phd_measurement* d_z = nullptr;
    phd_ackerman_noise* d_noise = nullptr;
    double* d_uniforms = nullptr; // n entries
    double* cdf = nullptr;        // n_global entries
    int* idx = nullptr;           // n_global entries
    // step report: status word, high-water marks, nEff and the resample decision live in ONE device block
    // (report[0..4]) so that one download fetches them all (phd_step_report, phd_state_snapshot)
    unsigned* report = nullptr;
    float* neff = nullptr;
    int* did = nullptr;
    float* state_pose = nullptr;
    int* state_argmax = nullptr;
    unsigned* status = nullptr;
    int* max_surv = nullptr;
    int* max_map = nullptr;
    int* d_tmp_int = nullptr; // n entries (selection / slot lists)
    int* h_plan = nullptr;    // pinned [3 n]: per slot the local parent (or -1) | the received row | the first slot that row fills: phd_global_resample_plan -> _end
    int* d_plan = nullptr;
    unsigned* ticket = nullptr; // arrival counter of the fused step (zero between launches)
    unsigned* gw_sync = nullptr;   // block form of the weights routine (n > 4096): counters of its grid-wide barriers (zero between launches)
    float* gw_part = nullptr;      // ... and one 8-word record per block of 256 weights
    bool fuse_enabled = true;
    // staging for AoS <-> SoA
    phd_gaussian2d* d_concat = nullptr;
    size_t concat_cap = 0;
    int* d_offsets = nullptr;
    int* d_sizes = nullptr;
    // inspection
    bool debug = false;
    float* dbg_surv = nullptr;
    int* dbg_u = nullptr;
    int* dbg_n = nullptr;
    int* dbg_nin = nullptr;
    unsigned long long* stamps = nullptr;
    bool want_stamps = false;
    int last_M = 0;

    // CPHD variant (filter_type = 1): per-particle log cardinality rows, double-buffered and indexed like the slabs
    bool cphd = false;
    int cn_len = 0;
    float* cn[2] = {nullptr, nullptr};
    float* d_lfact = nullptr;
    int lfact_len = 0;
    float2* cphd_scratch = nullptr; // [n_max][MM][MM]: rows of the ESF backward sweep (phd_kernels.hip, cphd_block)
    // migration plan of a global resample in flight (phd_global_resample_begin .. _end)
    std::vector<int32_t> plan_idx, plan_local_parent, plan_send, plan_recv_slots, plan_recv_rows, plan_row_first;
    void* send_buf = nullptr;
    size_t send_buf_bytes = 0;
    void* rows_target = nullptr; // caller-owned destination of phd_step_local_rows_dev's rows (in-place all-gather)
    bool rows_out = false;     // this launch writes its outputs into the export rows (phd_step_local_rows_dev)
    bool rows_pending = false; // ... and the step still has to be completed by phd_global_resample_gathered
    bool want_raw = false; // the update kernel also writes raw = logw + dlogw (multi-GPU step: no weights launch before the all-gather)
    GmWorkspace* gm = nullptr; // expected-map / gm_reduce workspace, created on first use
    int gm_rounds = 0;

    // pipelined state snapshot (phd_snapshot_capture / _send / _wait): two staging blocks on the device, their pinned mirrors, a
    // second stream for the downloads
    hipStream_t copy_stream = nullptr;
    hipEvent_t snap_ready[2] = {nullptr, nullptr}, snap_done[2] = {nullptr, nullptr};
    float* snap_dev[2] = {nullptr, nullptr};
    float* snap_host[2] = {nullptr, nullptr};
    int snap_state[2] = {0, 0};    // 0 idle, 1 captured, 2 sent
    int snap_n[2] = {0, 0};
    bool snap_idx[2] = {false, false};

    bool frozen = false;
    uint64_t seed = 0x5EED, counter = 0;
    const phd_pose* pose_for_update = nullptr; // set by a frozen predict

    bool timing = false;
    std::vector<TimedEvent> events;
    double t_ms[PHD_K_COUNT] = {0, 0, 0};
    int64_t t_n[PHD_K_COUNT] = {0, 0, 0};
};

// a shard of a sharded filter whose particle set has grown (particle shotgun, n_predict_particles > 1): every shard grows by the
// same factor, so the CURRENT global count and this shard's current offset follow from its own current count
static int shard_world(const phd_filter* f) { return f->n_base > 0 ? std::max(f->n_global / f->n_base, 1) : 1; }
static int ng_cur(const phd_filter* f) { return shard_world(f) * f->n; }
static int off_cur(const phd_filter* f) { return f->n_base > 0 ? (f->global_offset / f->n_base) * f->n : 0; }
// the first guest slab, counted from the start of the current buffer (the unit of the indirection)
static int guest_offset(const phd_filter* f) { return (2 - f->cur) * f->n_max; }
// may this resample take the copy-free form?  Only while the indirection is the identity (no resample since the last update:
// a guest slab an older indirection names must not be overwritten, and a peer must not find a guest index in this shard's
// view while this shard rewrites its guests), and, where a parent can be remote, only with a guest buffer
static bool copy_free_ok(const phd_filter* f, bool remote_possible)
{
    return f->copy_free && !f->parent_dirty && f->n == f->n_base && (!remote_possible || f->maps_g != nullptr);
}

static void fill_devcfg(const phd_slam_config& c, DevConfig& d)
{
    d.dt = c.dt;
    d.minRange = c.minRange; d.maxRange = c.maxRange; d.maxBearing = c.maxBearing;
    d.stdRange = c.stdRange; d.stdBearing = c.stdBearing;
    d.clutterDensity = c.clutterDensity; d.pd = c.pd;
    d.clutterRate = c.clutterRate;
    d.birthWeight = c.birthWeight; d.birthNoiseFactor = c.birthNoiseFactor;
    d.minFeatureWeight = c.minFeatureWeight; d.minSeparation = c.minSeparation;
    d.l = c.l; d.h = c.h; d.a = c.a; d.b = c.b;
    d.stdAlpha = c.stdAlpha; d.stdEncoder = c.stdEncoder;
    d.subdividePredict = c.subdividePredict > 0 ? c.subdividePredict : 1;
    d.distanceMetric = c.distanceMetric;
    d.labeledMeasurements = c.labeledMeasurements ? 1 : 0;
    d.particleOffset = 0;
}

static int check_supported(const phd_slam_config& c)
{
    // branches SURVEY.md §2 marks out of scope fail loudly instead of silently running something else
    if (c.featureModel != 0) return fail(PHD_ERR_UNSUPPORTED, "feature_model != 0 (dynamic/mixed features) is not supported");
    if (c.particleWeighting != 0) return fail(PHD_ERR_UNSUPPORTED, "particle_weighting != 0 is not supported");
    if (c.motionType != 1) return fail(PHD_ERR_UNSUPPORTED, "motion_type != 1 (Ackerman) is not supported");
    if (c.filterType != 0 && c.filterType != 1) return fail(PHD_ERR_INVALID_ARG, "filter_type must be 0 (PHD) or 1 (CPHD)");
    if (c.filterType == 1 && (c.maxCardinality < 1 || c.maxCardinality > 1023))
        return fail(PHD_ERR_INVALID_ARG, "CPHD: max_cardinality must be in 1..1023");
    if (c.distanceMetric != 0 && c.distanceMetric != 1) return fail(PHD_ERR_INVALID_ARG, "distance_metric must be 0 or 1");
    if (c.nPredictParticles < 1) return fail(PHD_ERR_INVALID_ARG, "n_predict_particles must be >= 1");
    return PHD_OK;
}

template <typename T>
static hipError_t dalloc(T** p, size_t n)
{
    return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
}

static int next_pow2(int x)
{
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

extern "C" int phd_create(const phd_slam_config* cfg, const phd_options* opt, phd_filter** out)
{
    if (!cfg || !out) return fail(PHD_ERR_INVALID_ARG, "phd_create: null argument");
    int rc = check_supported(*cfg);
    if (rc) return rc;
    phd_options o;
    memset(&o, 0, sizeof(o));
    if (opt) o = *opt;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PHD_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
    if (o.device < 0 || o.device >= ndev) return fail(PHD_ERR_INVALID_ARG, "phd_create: bad device ordinal");
    HIPCHK(hipSetDevice(o.device));

    phd_filter* f = new phd_filter();
    f->cfg = *cfg;
    fill_devcfg(*cfg, f->dcfg);
    f->device = o.device;
    f->n = o.n_particles > 0 ? o.n_particles : cfg->n_particles;
    f->n_base = f->n;
    // particle "shotgun" (n_predict_particles = k > 1, src/phdfilter.cu:1185-1238): every predict multiplies
    // the count by k until it exceeds 5*n_particles and the resample trigger of src/main.cpp:1286 fires
    f->n_max = cfg->nPredictParticles > 1 ? 5 * f->n * cfg->nPredictParticles : f->n;
    f->cap = o.map_capacity > 0 ? o.map_capacity : 256;
    f->MM = o.max_measurements > 0 ? std::min(o.max_measurements, PHD_MAX_MEASUREMENTS) : PHD_MAX_MEASUREMENTS;
    f->M_limit = f->MM;
    // The update kernel has instantiations with an LDS layout compiled in (phd_kernels.hip, LAYOUT: 1024 survivor slots / map capacity
    // 512 / 64 measurements and 512 / 128 / 32) — worth +3 % (PHD) to +25 % (CPHD) on real scans.  The layout is a property of the
    // LIBRARY, not of a bench script (round 6): a filter with one of those map capacities and the default survivor
    // capacity gets its measurement capacity rounded UP to the layout's — LDS for a few more measurements; scans are still clamped to
    // the caller's max_measurements (M_limit).  (Whatever PHD_LAYOUT says: that switch picks instantiations, not capacities — the CPHD
    // block chooses some of its forms by the measurement capacity, and a filter must not change its arithmetic with a debugging switch.)
    if (o.survivor_capacity <= 0) {
        if (f->cap == 512 && f->MM < 64) f->MM = 64;
        else if (f->cap == 128 && f->MM < 32) f->MM = 32;
    }
    f->n_global = o.global_particles > 0 ? o.global_particles : f->n;
    f->global_offset = o.global_offset;
    f->dcfg.particleOffset = o.global_offset;
    if (f->n <= 0) { delete f; return fail(PHD_ERR_INVALID_ARG, "phd_create: n_particles <= 0"); }
    if (f->cap > 65535) { delete f; return fail(PHD_ERR_INVALID_ARG, "phd_create: map_capacity > 65535"); }
    int S = o.survivor_capacity > 0 ? o.survivor_capacity : (f->cap + 8 * f->MM);
    S = next_pow2(std::max(S, 256)); // the register sorts own 256*E slots
    if (S > 2048) S = 2048; // register-staged permutation in merge_in_lds handles <= 2048
    f->S_cap = S;
    // a request beyond what LDS holds: the spill list (every particle can then carry up to survivor_capacity survivors;
    // those with more than 2048 take the global-memory merge of phd_spill.h — slower, but no PHD_ERR_CAPACITY)
    if (o.survivor_capacity > 2048) f->spill_cap = std::min((o.survivor_capacity + 511) / 512 * 512, 32768);
    f->lds_bytes = update_lds_bytes(f->S_cap, f->cap, f->MM);
    if (cfg->filterType == 1) {
        f->cphd = true;
        f->cn_len = cfg->maxCardinality + 1;
        f->lfact_len = std::max(f->cn_len, f->MM + 1) + 1;
        f->lds_bytes += cphd_lds_bytes(f->S_cap, f->cn_len, f->MM);
    }
    // a launch can use 160 KiB minus what the kernel's instantiations declare statically (the fused ones carry the
    // weights routine's arrays): refuse here what launch_update_merge could not launch
    if (f->lds_bytes + update_static_lds_bytes() > 160 * 1024) {
        delete f;
        return fail(PHD_ERR_CAPACITY, "phd_create: map_capacity/survivor_capacity need more than 160 KiB of LDS");
    }
    // the build this filter runs, for its whole life (PHD_UPDATE_BUILD=2 / 3 overrides: the tests compare the two builds bit for bit)
    f->three_per_cu = update_takes_three_per_cu(f->cphd, f->spill_cap != 0, f->lds_bytes, f->n_base);
    if (const char* e = getenv("PHD_LAYOUT")) f->any_layout = e[0] == '0';
    if (const char* e = getenv("PHD_UPDATE_BUILD")) {
        if (e[0] == '2') f->three_per_cu = false;
        else if (e[0] == '3' && !f->spill_cap && 3 * (f->lds_bytes + 1024) <= 160 * 1024) f->three_per_cu = true;
    }
    if (o.stream) {
        f->stream = (hipStream_t)o.stream;
    } else {
        if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) {
            delete f;
            return fail(PHD_ERR_HIP, "hipStreamCreate failed");
        }
        f->own_stream = true;
    }
    const size_t slab = (size_t)f->n_max * 6 * f->cap;
    // buffers that hold the GLOBAL particle set of a sharded filter: with the particle shotgun the global set grows like the shard
    const size_t gmax = std::max<size_t>(std::max(f->n_max, f->n_global), (size_t)shard_world(f) * f->n_max);
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess && r != hipSuccess) e = r; };
    {
        const int nbuf = shard_world(f) > 1 ? 3 : 2;            // (a guest buffer only where a parent can be remote)
        A(dalloc(&f->maps_arena, nbuf * slab)); A(dalloc(&f->counts_arena, (size_t)nbuf * f->n_max));
        if (e == hipSuccess) {
            for (int k = 0; k < 2; ++k) { f->maps[k] = f->maps_arena + k * slab; f->counts[k] = f->counts_arena + (size_t)k * f->n_max; }
            if (nbuf == 3) { f->maps_g = f->maps_arena + 2 * slab; f->counts_g = f->counts_arena + 2 * (size_t)f->n_max; }
        }
        if (const char* ev = getenv("PHD_COPY_FREE")) f->copy_free = ev[0] != '0';
    }
    for (int k = 0; k < 3; ++k) { A(dalloc(&f->parent[k], f->n_max)); A(dalloc(&f->pose[k], f->n_max)); }
    A(dalloc(&f->logw, f->n_max)); A(dalloc(&f->logw_alt, f->n_max)); A(dalloc(&f->logw_scratch, gmax));
    A(dalloc(&f->logw_raw_own, f->n_max)); A(dalloc(&f->dlogw, f->n_max));
    f->logw_raw = f->logw_raw_own;
    A(dalloc(&f->d_z, f->MM)); A(dalloc(&f->d_noise, f->n_max));
    A(dalloc(&f->d_uniforms, gmax));
    A(dalloc(&f->cdf, gmax));
    A(dalloc(&f->idx, gmax));
    A(dalloc(&f->report, 8));
    f->status = f->report; f->max_surv = (int*)f->report + 1; f->max_map = (int*)f->report + 2;
    f->neff = (float*)f->report + 3; f->did = (int*)f->report + 4;
    A(dalloc(&f->state_pose, 6)); A(dalloc(&f->state_argmax, 1));
    A(dalloc(&f->d_tmp_int, f->n_max));
    A(dalloc(&f->d_plan, 3 * (size_t)f->n_max));
    A(hipHostMalloc((void**)&f->h_plan, 3 * (size_t)f->n_max * sizeof(int)));
    A(dalloc(&f->ticket, 1));
    A(dalloc(&f->gw_sync, 4)); A(dalloc(&f->gw_part, 8 * ((gmax + 255) / 256 + 1)));
    A(dalloc(&f->d_offsets, f->n_max + 1)); A(dalloc(&f->d_sizes, gmax));
    if (f->spill_cap) {
        A(dalloc(&f->spill_rec, (size_t)f->n_max * 2 * f->spill_cap * 8));
        A(dalloc(&f->spill_meta, (size_t)f->n_max * 8));
        A(dalloc(&f->spill_out, (size_t)f->n_max * f->cap));
        A(dalloc(&f->spill_acc, (size_t)f->n_max * f->cap * 8));
        A(dalloc(&f->spill_tmp, (size_t)f->n_max * f->spill_cap));
    }
    if (f->cphd) {
        const size_t rows = (size_t)f->n_max * f->cn_len;
        A(dalloc(&f->cn_arena, (f->maps_g ? 3 : 2) * rows));
        if (e == hipSuccess) { f->cn[0] = f->cn_arena; f->cn[1] = f->cn_arena + rows; if (f->maps_g) f->cn_g = f->cn_arena + 2 * rows; }
        A(dalloc(&f->d_lfact, f->lfact_len));
        A(dalloc(&f->cphd_scratch, (size_t)f->n_max * f->MM * f->MM));
    }
    if (e != hipSuccess) {
        phd_destroy(f);
        return fail(PHD_ERR_HIP, std::string("device allocation failed: ") + hipGetErrorString(e));
    }
    hipMemsetAsync(f->maps_arena, 0, (f->maps_g ? 3 : 2) * slab * sizeof(float), f->stream);
    hipMemsetAsync(f->counts_arena, 0, (f->maps_g ? 3 : 2) * (size_t)f->n_max * sizeof(int), f->stream);
    hipMemsetAsync(f->report, 0, 8 * 4, f->stream);
    hipMemsetAsync(f->ticket, 0, 4, f->stream);
    hipMemsetAsync(f->gw_sync, 0, 16, f->stream);
    for (int k = 0; k < 3; ++k) launch_iota(f->parent[k], f->n_max, f->stream);
    if (f->cphd) {
        // uniform cardinality -log(maxCardinality+1) (src/main.cpp:1142); log factorials by the reference's
        // recurrence (initCphdConstants, src/phdfilter.cu.bak:421-425)
        for (int k = 0; k < 2; ++k) launch_fill(f->cn[k], -logf((float)f->cn_len), f->n_max * f->cn_len, f->stream);
        std::vector<float> lf(f->lfact_len);
        lf[0] = 0.f;
        for (int k = 1; k < f->lfact_len; ++k) lf[k] = lf[k - 1] + logf((float)k);
        hipMemcpyAsync(f->d_lfact, lf.data(), lf.size() * sizeof(float), hipMemcpyHostToDevice, f->stream);
        hipStreamSynchronize(f->stream);
    }
    // initial particles: cfg pose, weights -log N (src/main.cpp:1130-1145)
    std::vector<phd_pose> p0(f->n);
    std::vector<float> w0(f->n, -logf((float)f->n_global));
    for (auto& q : p0) { q.px = cfg->x0; q.py = cfg->y0; q.ptheta = cfg->yaw0; q.vx = cfg->vx0; q.vy = cfg->vy0; q.vtheta = cfg->vyaw0; }
    hipMemcpyAsync(f->pose[0], p0.data(), f->n * sizeof(phd_pose), hipMemcpyHostToDevice, f->stream);
    hipMemcpyAsync(f->logw, w0.data(), f->n * sizeof(float), hipMemcpyHostToDevice, f->stream);
    if (hipStreamSynchronize(f->stream) != hipSuccess) {
        phd_destroy(f);
        return fail(PHD_ERR_HIP, "initialisation failed");
    }
    *out = f;
    return PHD_OK;
}

extern "C" int phd_destroy(phd_filter* f)
{
    if (!f) return PHD_OK;
    hipSetDevice(f->device);
    if (f->stream) hipStreamSynchronize(f->stream);
    for (auto& ev : f->events) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    hipFree(f->maps_arena); hipFree(f->counts_arena); hipFree(f->cn_arena);
    for (int k = 0; k < 3; ++k) { hipFree(f->parent[k]); hipFree(f->pose[k]); }
    hipFree(f->logw); hipFree(f->logw_alt); hipFree(f->logw_scratch); hipFree(f->logw_raw_own); hipFree(f->dlogw);
    hipFree(f->d_z); hipFree(f->d_noise); hipFree(f->d_uniforms); hipFree(f->cdf); hipFree(f->idx);
    hipFree(f->report); hipFree(f->state_pose); hipFree(f->state_argmax);
    hipFree(f->d_tmp_int); hipFree(f->ticket); hipFree(f->gw_sync); hipFree(f->gw_part); hipFree(f->d_plan);
    if (f->h_plan) hipHostFree(f->h_plan);
    hipFree(f->d_concat); hipFree(f->d_offsets); hipFree(f->d_sizes);
    hipFree(f->d_lfact); hipFree(f->cphd_scratch); hipFree(f->send_buf);
    hipFree(f->dbg_surv); hipFree(f->dbg_u); hipFree(f->dbg_n); hipFree(f->dbg_nin); hipFree(f->stamps);
    hipFree(f->spill_rec); hipFree(f->spill_meta); hipFree(f->spill_out); hipFree(f->spill_acc); hipFree(f->spill_tmp);
    gm_workspace_destroy(f->gm);
    if (f->copy_stream) { hipStreamSynchronize(f->copy_stream); hipStreamDestroy(f->copy_stream); }
    for (int k = 0; k < 2; ++k) {
        if (f->snap_ready[k]) hipEventDestroy(f->snap_ready[k]);
        if (f->snap_done[k]) hipEventDestroy(f->snap_done[k]);
        hipFree(f->snap_dev[k]);
        if (f->snap_host[k]) hipHostFree(f->snap_host[k]);
    }
    if (f->own_stream && f->stream) hipStreamDestroy(f->stream);
    delete f;
    return PHD_OK;
}

#define CHECK_F0(f)                                                      \
    do {                                                                 \
        if (!(f)) return fail(PHD_ERR_INVALID_ARG, "null filter handle"); \
        HIPCHK(hipSetDevice((f)->device));                               \
    } while (0)
// every entry point but the ones that complete or only observe a phd_step_local_rows_dev step
#define CHECK_F(f)                                                       \
    do {                                                                 \
        CHECK_F0(f);                                                     \
        if ((f)->rows_pending)                                           \
            return fail(PHD_ERR_INVALID_ARG, "phd_step_local_rows_dev is pending: complete the step with phd_global_resample_gathered"); \
    } while (0)

extern "C" int phd_set_config(phd_filter* f, const phd_slam_config* cfg)
{
    CHECK_F(f);
    if (!cfg) return fail(PHD_ERR_INVALID_ARG, "null config");
    int rc = check_supported(*cfg);
    if (rc) return rc;
    f->cfg = *cfg;
    fill_devcfg(*cfg, f->dcfg); // kernel argument from now on; nothing to upload
    f->dcfg.particleOffset = f->global_offset;
    return PHD_OK;
}

extern "C" int phd_seed(phd_filter* f, uint64_t seed)
{
    CHECK_F(f);
    f->seed = seed;
    f->counter = 0;
    return PHD_OK;
}

extern "C" int phd_n_particles(const phd_filter* f) { return f ? f->n : PHD_ERR_INVALID_ARG; }
extern "C" int phd_map_capacity(const phd_filter* f) { return f ? f->cap : PHD_ERR_INVALID_ARG; }
extern "C" void* phd_stream(phd_filter* f) { return f ? (void*)f->stream : nullptr; }

extern "C" int phd_sync(phd_filter* f)
{
    CHECK_F0(f);
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// timing helpers
// ---------------------------------------------------------------------------------------------
static void t_begin(phd_filter* f, int kind)
{
    if (!f->timing) return;
    TimedEvent ev;
    ev.kind = kind;
    hipEventCreate(&ev.a);
    hipEventCreate(&ev.b);
    hipEventRecord(ev.a, f->stream);
    f->events.push_back(ev);
}
static void t_end(phd_filter* f)
{
    if (!f->timing) return;
    hipEventRecord(f->events.back().b, f->stream);
}
static void t_collect(phd_filter* f)
{
    for (auto& ev : f->events) {
        float ms = 0.f;
        hipEventSynchronize(ev.b);
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { f->t_ms[ev.kind] += ms; f->t_n[ev.kind] += 1; }
        hipEventDestroy(ev.a);
        hipEventDestroy(ev.b);
    }
    f->events.clear();
}

extern "C" int phd_timing_enable(phd_filter* f, int enable)
{
    CHECK_F0(f);
    if (!enable) t_collect(f);
    f->timing = enable != 0;
    return PHD_OK;
}
extern "C" int phd_timing_reset(phd_filter* f)
{
    CHECK_F0(f);
    t_collect(f);
    for (int k = 0; k < PHD_K_COUNT; ++k) { f->t_ms[k] = 0; f->t_n[k] = 0; }
    return PHD_OK;
}
extern "C" int phd_timing_read(phd_filter* f, double* ms_total, int64_t* launches)
{
    CHECK_F0(f);
    t_collect(f);
    for (int k = 0; k < PHD_K_COUNT; ++k) {
        if (ms_total) ms_total[k] = f->t_ms[k];
        if (launches) launches[k] = f->t_n[k];
    }
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// state transfer
// ---------------------------------------------------------------------------------------------
static int materialize_parents(phd_filter* f)
{
    // resolve the map indirection left by a resample: maps[cur^1][p] = maps[cur][parent[p]]
    if (!f->parent_dirty) return PHD_OK;
    HIPCHK(launch_gather_maps(f->maps[f->cur], f->counts[f->cur], f->parent[f->pcur], nullptr, f->maps[f->cur ^ 1],
                              f->counts[f->cur ^ 1], nullptr, nullptr, f->cap, f->n, f->stream));
    if (f->cphd)
        HIPCHK(launch_copy_rows(f->cn[f->cur], f->cn_len, nullptr, f->parent[f->pcur], f->cn[f->cur ^ 1], f->cn_len, nullptr,
                                f->cn_len, f->n, f->stream));
    f->cur ^= 1;
    HIPCHK(launch_iota(f->parent[f->pcur], f->n, f->stream));
    f->parent_dirty = false;
    return PHD_OK;
}

extern "C" int phd_set_particles(phd_filter* f, const phd_pose* poses, const float* log_weights, int n)
{
    CHECK_F(f);
    if (n != f->n) return fail(PHD_ERR_INVALID_ARG, "phd_set_particles: n != n_particles");
    if (poses) HIPCHK(hipMemcpyAsync(f->pose[f->pose_cur], poses, n * sizeof(phd_pose), hipMemcpyHostToDevice, f->stream));
    if (log_weights) HIPCHK(hipMemcpyAsync(f->logw, log_weights, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_get_particles(phd_filter* f, phd_pose* poses_out, float* log_weights_out)
{
    CHECK_F(f);
    if (poses_out) HIPCHK(hipMemcpyAsync(poses_out, f->pose[f->pose_cur], f->n * sizeof(phd_pose), hipMemcpyDeviceToHost, f->stream));
    if (log_weights_out) HIPCHK(hipMemcpyAsync(log_weights_out, f->logw, f->n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

static int ensure_concat(phd_filter* f, size_t n)
{
    if (n <= f->concat_cap) return PHD_OK;
    HIPCHK(hipStreamSynchronize(f->stream));
    if (f->d_concat) hipFree(f->d_concat);
    f->d_concat = nullptr;
    f->concat_cap = 0;
    size_t want = std::max(n, (size_t)f->n * 16);
    HIPCHK(dalloc(&f->d_concat, want));
    f->concat_cap = want;
    return PHD_OK;
}

extern "C" int phd_set_maps(phd_filter* f, const phd_gaussian2d* concat, const int32_t* sizes)
{
    CHECK_F(f);
    if (!sizes) return fail(PHD_ERR_INVALID_ARG, "phd_set_maps: null sizes");
    if (f->cphd) { // the cardinality rows share the map indirection this call resets
        int rc = materialize_parents(f);
        if (rc) return rc;
    }
    std::vector<int> off(f->n + 1, 0);
    for (int p = 0; p < f->n; ++p) {
        if (sizes[p] < 0 || sizes[p] > f->cap)
            return fail(PHD_ERR_CAPACITY, "phd_set_maps: a map exceeds map_capacity");
        off[p + 1] = off[p] + sizes[p];
    }
    const size_t total = off[f->n];
    if (total && !concat) return fail(PHD_ERR_INVALID_ARG, "phd_set_maps: null maps");
    int rc = ensure_concat(f, total);
    if (rc) return rc;
    if (total) HIPCHK(hipMemcpyAsync(f->d_concat, concat, total * sizeof(phd_gaussian2d), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_offsets, off.data(), (f->n + 1) * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->counts[f->cur], sizes, f->n * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(launch_pack_maps(f->d_concat, f->d_offsets, f->counts[f->cur], f->maps[f->cur], f->cap, f->n, f->stream));
    HIPCHK(launch_iota(f->parent[f->pcur], f->n, f->stream));
    f->parent_dirty = false;
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_get_map_sizes(phd_filter* f, int32_t* sizes_out)
{
    CHECK_F(f);
    if (!sizes_out) return fail(PHD_ERR_INVALID_ARG, "null output");
    // a parent may be any slab written while the particle set was larger (shotgun): all n_max counts
    // (and, after a shard's copy-free resample, a slab of the guest buffer: everything from the current buffer to the end)
    const size_t reach = (size_t)((f->maps_g ? 3 : 2) - f->cur) * f->n_max;
    std::vector<int> cnt(reach), par(f->n);
    HIPCHK(hipMemcpyAsync(cnt.data(), f->counts[f->cur], reach * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(par.data(), f->parent[f->pcur], f->n * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    for (int p = 0; p < f->n; ++p) sizes_out[p] = cnt[par[p]];
    return PHD_OK;
}

extern "C" int phd_get_maps(phd_filter* f, phd_gaussian2d* concat_out, size_t concat_capacity, int32_t* sizes_out)
{
    CHECK_F(f);
    std::vector<int32_t> sizes(f->n);
    int rc = phd_get_map_sizes(f, sizes.data());
    if (rc) return rc;
    std::vector<int> off(f->n + 1, 0);
    for (int p = 0; p < f->n; ++p) off[p + 1] = off[p] + sizes[p];
    const size_t total = off[f->n];
    if (sizes_out) memcpy(sizes_out, sizes.data(), f->n * sizeof(int32_t));
    if (!concat_out) return PHD_OK;
    if (total > concat_capacity) return fail(PHD_ERR_CAPACITY, "phd_get_maps: output buffer too small");
    rc = ensure_concat(f, total);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(f->d_offsets, off.data(), (f->n + 1) * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(launch_unpack_maps(f->maps[f->cur], f->parent[f->pcur], f->d_offsets, f->counts[f->cur], f->d_concat, f->cap,
                              f->n, f->stream));
    if (total) HIPCHK(hipMemcpyAsync(concat_out, f->d_concat, total * sizeof(phd_gaussian2d), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_set_map(phd_filter* f, int particle, const phd_gaussian2d* g, int n)
{
    CHECK_F(f);
    if (particle < 0 || particle >= f->n) return fail(PHD_ERR_INVALID_ARG, "phd_set_map: bad particle index");
    if (n < 0 || n > f->cap) return fail(PHD_ERR_CAPACITY, "phd_set_map: map exceeds map_capacity");
    int rc = materialize_parents(f);
    if (rc) return rc;
    std::vector<float> slab((size_t)6 * f->cap, 0.f);
    for (int i = 0; i < n; ++i) {
        slab[0 * f->cap + i] = g[i].weight;
        slab[1 * f->cap + i] = g[i].mean[0];
        slab[2 * f->cap + i] = g[i].mean[1];
        slab[3 * f->cap + i] = g[i].cov[0];
        slab[4 * f->cap + i] = (g[i].cov[1] + g[i].cov[2]) * 0.5f;
        slab[5 * f->cap + i] = g[i].cov[3];
    }
    HIPCHK(hipMemcpyAsync(f->maps[f->cur] + (size_t)particle * 6 * f->cap, slab.data(), slab.size() * sizeof(float),
                          hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->counts[f->cur] + particle, &n, sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_get_map(phd_filter* f, int particle, phd_gaussian2d* out, int capacity, int32_t* n_out)
{
    CHECK_F(f);
    if (particle < 0 || particle >= f->n) return fail(PHD_ERR_INVALID_ARG, "phd_get_map: bad particle index");
    int src = particle;
    HIPCHK(hipMemcpyAsync(&src, f->parent[f->pcur] + particle, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, f->counts[f->cur] + src, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    std::vector<float> slab((size_t)6 * f->cap);
    HIPCHK(hipMemcpyAsync(slab.data(), f->maps[f->cur] + (size_t)src * 6 * f->cap, slab.size() * sizeof(float),
                          hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (n_out) *n_out = n;
    if (!out) return PHD_OK;
    if (n > capacity) return fail(PHD_ERR_CAPACITY, "phd_get_map: output buffer too small");
    for (int i = 0; i < n; ++i) {
        out[i].weight = slab[0 * f->cap + i];
        out[i].mean[0] = slab[1 * f->cap + i];
        out[i].mean[1] = slab[2 * f->cap + i];
        out[i].cov[0] = slab[3 * f->cap + i];
        out[i].cov[1] = slab[4 * f->cap + i];
        out[i].cov[2] = slab[4 * f->cap + i];
        out[i].cov[3] = slab[5 * f->cap + i];
    }
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------------------------
static int do_predict(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise)
{
    const int k = f->cfg.nPredictParticles;
    if (k > 1) {
        // shotgun: k predicted particles per prior particle (prior index = idx / k, src/phdfilter.cu:797);
        // maps are shared through the parent indirection, weights lose log k (:1213)
        if (f->frozen) return fail(PHD_ERR_UNSUPPORTED, "n_predict_particles > 1 is not supported while frozen");
        if ((long long)f->n * k > f->n_max)
            return fail(PHD_ERR_CAPACITY, "particle count would exceed 5*n_particles*n_predict_particles: resample first (src/main.cpp:1286)");
        const int pnext = (f->pose_cur + 1) % 3;
        DevConfig dc = f->dcfg;
        dc.particleOffset = off_cur(f) * k;                 // a shard's first predicted particle in the GLOBAL predicted ordering
        t_begin(f, PHD_K_PREDICT);
        HIPCHK(launch_predict_shotgun(f->pose[f->pose_cur], f->pose[pnext], f->n * k, k, u, d_noise, f->seed, f->counter,
                                      dc, f->parent[f->pcur], f->parent[f->pcur ^ 1], f->logw, f->logw_alt, f->stream));
        t_end(f);
        f->counter++;
        f->pose_cur = pnext;
        f->pcur ^= 1;
        std::swap(f->logw, f->logw_alt);
        f->n *= k;
        f->parent_dirty = true;
        f->pose_for_update = nullptr;
        return PHD_OK;
    }
    const phd_pose* in = f->pose[f->pose_cur];
    phd_pose* out = f->frozen ? f->pose[(f->pose_cur + 1) % 3] : f->pose[f->pose_cur];
    t_begin(f, PHD_K_PREDICT);
    HIPCHK(launch_predict(in, out, f->n, u, d_noise, f->seed, f->counter, f->dcfg, f->stream));
    t_end(f);
    f->counter++;
    f->pose_for_update = f->frozen ? out : nullptr;
    return PHD_OK;
}

extern "C" int phd_predict_ackerman_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise)
{
    CHECK_F(f);
    return do_predict(f, u, d_noise);
}

extern "C" int phd_predict_ackerman(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* noise)
{
    CHECK_F(f);
    if (noise) {
        const int k = f->cfg.nPredictParticles > 1 ? f->cfg.nPredictParticles : 1;
        if ((long long)f->n * k > f->n_max) return fail(PHD_ERR_CAPACITY, "particle count would exceed its maximum: resample first");
        HIPCHK(hipMemcpyAsync(f->d_noise, noise, (size_t)f->n * k * sizeof(phd_ackerman_noise), hipMemcpyHostToDevice, f->stream));
        return do_predict(f, u, f->d_noise);
    }
    return do_predict(f, u, nullptr);
}

static int ensure_debug(phd_filter* f)
{
    if (f->dbg_surv) return PHD_OK;
    const int dc = f->spill_cap ? f->spill_cap : f->S_cap;
    HIPCHK(dalloc(&f->dbg_surv, (size_t)f->n_max * 6 * dc));
    HIPCHK(dalloc(&f->dbg_u, (size_t)f->n_max * dc));
    HIPCHK(dalloc(&f->dbg_n, f->n_max));
    HIPCHK(dalloc(&f->dbg_nin, f->n_max));
    return PHD_OK;
}

struct FusedPredict {
    phd_ackerman_control u;
    const phd_ackerman_noise* d_noise;
};

struct FusedWeights {
    int mode;
    double u0;
};
static void build_weight_args(phd_filter* f, int mode, const double* d_uniforms, int n_uniforms, double u0,
                              WeightArgs& w, int& free_pose);
// every launch of the weights routine goes through here: the block form (n > 4096) needs the filter's barrier counters and block
// records, and reports a time-out in the filter's status word
static hipError_t launch_weights_f(phd_filter* f, WeightArgs& w)
{
    w.gsync = f->gw_sync;
    w.gpart = f->gw_part;
    return launch_weights(w, f->stream, f->status);
}
static int interpret_report(phd_filter* f, const unsigned raw[8], phd_step_report* out);
static void commit_weights(phd_filter* f, int mode, int free_pose);

#ifdef PHD_EXP_TRACE
static unsigned long long* g_exp_trace = nullptr;
extern "C" int phd_exp_trace_read(unsigned long long* out, int n_rows)
{
    if (!g_exp_trace) return -1;
    hipDeviceSynchronize();
    return hipMemcpy(out, g_exp_trace, sizeof(unsigned long long) * 8 * n_rows, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
static int do_update_merge(phd_filter* f, const phd_measurement* d_z, int M, const FusedPredict* fp = nullptr,
                           const FusedWeights* fw = nullptr)
{
    UpdateArgs a;
    memset(&a, 0, sizeof(a));
    a.map_in = f->maps[f->cur];
    a.count_in = f->counts[f->cur];
    a.map_out = f->maps[f->cur ^ 1];
    a.count_out = f->counts[f->cur ^ 1];
    if (f->rows_out) { // the outputs land in the export rows (phd_step_local_rows_dev)
        a.map_out = (float*)(f->rows_target ? f->rows_target : f->send_buf) + 8;
        a.out_stride = (unsigned)(phd_particle_pack_bytes(f) / 4);
    }
    a.parent = f->parent[f->pcur];
    a.parent_reset = f->frozen ? nullptr : f->parent[f->pcur];
    a.pose = f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur];
    if (fp) {
        // vehicle predict fused into the update kernel: reads the live pose, writes the predicted one
        a.pose = f->pose[f->pose_cur];
        a.pose_out = f->frozen ? f->pose[(f->pose_cur + 1) % 3] : f->pose[f->pose_cur];
        a.do_predict = 1;
        a.control = fp->u;
        a.noise = fp->d_noise;
        a.seed = f->seed;
        a.counter = f->counter++;
        f->pose_for_update = f->frozen ? a.pose_out : nullptr; // what the weights kernel gathers from
    }
    a.z = d_z;
    a.dlogw = f->dlogw;
    a.logw_in = f->logw;
    a.raw_out = f->want_raw ? f->logw_raw : nullptr;
    a.M = M;
    a.MM = f->MM;
    a.cap = f->cap;
    a.S_cap = f->S_cap;
    if (f->debug) {
        int rc = ensure_debug(f);
        if (rc) return rc;
        a.dbg_surv = f->dbg_surv; a.dbg_u = f->dbg_u; a.dbg_n = f->dbg_n; a.dbg_nin = f->dbg_nin;
        a.dbg_cap = f->spill_cap ? f->spill_cap : f->S_cap;
    }
    if (f->want_stamps) {
        if (!f->stamps) {
            HIPCHK(dalloc(&f->stamps, (size_t)f->n_max * PHD_STAMP_ROW + 8));
            HIPCHK(hipMemsetAsync(f->stamps, 0, ((size_t)f->n_max * PHD_STAMP_ROW + 8) * 8, f->stream));
        }
        a.stamps = f->stamps;
    }
    a.status = f->status;
    a.max_surv = f->max_surv;
    a.max_map = f->max_map;
    if (f->cphd) {
        a.cphd = 1;
        a.cn_in = f->cn[f->cur];
        a.cn_out = f->cn[f->cur ^ 1];
        if (f->rows_out) a.cn_out = (float*)(f->rows_target ? f->rows_target : f->send_buf) + 8 + 6 * f->cap;
        a.cn_len = f->cn_len;
        a.lfact = f->d_lfact;
        a.lfact_len = f->lfact_len;
        a.cphd_scratch = f->cphd_scratch;
    }
    if (f->spill_cap) {
        a.spill_rec = f->spill_rec; a.spill_meta = f->spill_meta; a.spill_out = f->spill_out; a.spill_cap = f->spill_cap;
        a.spill_acc = f->spill_acc; a.spill_tmp = f->spill_tmp;
    }
    a.cfg = f->dcfg;
    int free_pose = 0;
    if (fw) {
        build_weight_args(f, fw->mode, nullptr, 1, fw->u0, a.wa, free_pose);
        a.fuse_weights = 1;
        a.ticket = f->ticket;
#ifdef PHD_EXP_TRACE
        {
            static unsigned long long* g_trace = nullptr;
            if (f->n + 1 > 70000) return fail(PHD_ERR_CAPACITY, "PHD_EXP_TRACE: the trace buffer holds 70000 workgroups");
            if (!g_trace) hipMalloc(&g_trace, sizeof(unsigned long long) * 8 * 70000);
            a.trace = g_trace;
            g_exp_trace = g_trace;
        }
#endif
    }
    t_begin(f, PHD_K_UPDATE_MERGE);
    HIPCHK(launch_update_merge(a, f->n, f->lds_bytes, f->stream, f->three_per_cu, f->any_layout));
    f->last_update_fn = update_instantiation(a, f->n, f->three_per_cu, f->any_layout);
    HIPCHK(launch_merge_spill(a, f->n, f->stream));   // (a no-op without a spill list) particles whose survivors outgrew LDS
    t_end(f);
    f->last_M = M;
    if (!f->frozen) {
        f->cur ^= 1;
        f->parent_dirty = false;
    }
    if (fw) commit_weights(f, fw->mode, free_pose);
    return PHD_OK;
}

// weights kernel on the local shard
// arguments of the weights / nEff / resample routine on the local shard; free_pose = the pose
// buffer the commit gathers into
static void build_weight_args(phd_filter* f, int mode, const double* d_uniforms, int n_uniforms, double u0,
                              WeightArgs& w, int& free_pose)
{
    memset(&w, 0, sizeof(w));
    w.logw_in = f->logw;
    w.logw = f->frozen ? f->logw_scratch : f->logw;
    w.dlogw = f->dlogw;
    w.raw_out = (mode & WM_NORMALIZE) ? nullptr : ((mode & WM_ACCUMULATE) ? f->logw_raw : nullptr);
    w.n = f->n;
    w.n_new = (f->n == f->n_base || f->n_global != f->n_base) ? f->n : f->n_base; // resample back to n_particles (src/main.cpp:1289)
    w.mode = mode;
    w.resample_thresh = f->cfg.resampleThresh;
    w.uniforms = d_uniforms ? d_uniforms : f->d_uniforms;
    w.n_uniforms = n_uniforms;
    w.u0 = u0;
    w.cdf = f->cdf;
    w.idx_out = f->idx;
    w.neff_out = f->neff;
    w.did_resample = f->did;
    const phd_pose* pin = f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur];
    free_pose = 0;
    while (f->pose[free_pose] == pin || free_pose == f->pose_cur) free_pose++;
    w.pose_in = pin;
    w.pose_out = f->pose[free_pose];
    w.parent_in = f->parent[f->pcur];
    w.parent_out = f->frozen ? f->parent[2] : f->parent[f->pcur ^ 1];
    w.n_weight_norm = f->n_global;
    if (f->want_stamps && f->stamps) w.wstamps = f->stamps + (size_t)f->n * PHD_STAMP_ROW;
    w.gsync = f->gw_sync;
    w.gpart = f->gw_part;
}

static void commit_weights(phd_filter* f, int mode, int free_pose)
{
    if ((mode & WM_COMMIT) && !f->frozen) {
        f->pose_cur = free_pose;
        f->pcur ^= 1;
        f->parent_dirty = true; // conservatively: parents may be non-identity now
    }
    f->pose_for_update = nullptr;
}

static int do_weights(phd_filter* f, int mode, const double* d_uniforms, int n_uniforms, double u0 = 0.0)
{
    WeightArgs w;
    int free_pose = 0;
    build_weight_args(f, mode, d_uniforms, n_uniforms, u0, w, free_pose);
    // large sets: the routine's searches run on several workgroups that do not wait for each other, so the weights are
    // written out of place (launch_weights) and the buffers swapped
    const bool out_of_place = !f->frozen && f->n > 1024 && f->n <= weights_grid_min_particles() && (mode & (WM_RESAMPLE_FORCE | WM_RESAMPLE_AUTO));
    if (out_of_place) w.logw = f->logw_alt;
    t_begin(f, PHD_K_WEIGHTS);
    HIPCHK(launch_weights_f(f, w));
    t_end(f);
    if (out_of_place) std::swap(f->logw, f->logw_alt);
    commit_weights(f, mode, free_pose);
    return PHD_OK;
}

// update + prune + merge with the weights routine fused into the kernel's tail (the last workgroup
// to finish runs it): one launch per step.  Used for small particle counts, where the second
// launch's fixed cost is a measurable share of the step.
static bool can_fuse(const phd_filter* f)
{
    if (!f->fuse_enabled || f->want_stamps || f->n != f->n_base || f->n > update_fuse_max_particles()) return false;
    // up to 4096 particles the tail is ONE workgroup with the fixed-point CDF in LDS; above, the block form on several
    // workgroups (phd_weights.h: weights_grid_body), which needs the block ends of the CDF there
    if (f->n <= weights_grid_min_particles()) return (size_t)f->n * 8 <= f->lds_bytes;
    // (PHD filters without a spill list: the instantiations that carry it, phd_kernels.hip)
    return !f->cphd && !f->spill_cap && weights_grid_lds_bytes(f->n) <= f->lds_bytes;
}

extern "C" int phd_update_residency(phd_filter* f, int32_t* workgroups_per_cu_out, uint64_t* lds_bytes_out)
{
    CHECK_F(f);
    HIPCHK(hipSetDevice(f->device));
    UpdateArgs a;
    memset(&a, 0, sizeof(a));
    a.cphd = f->cphd ? 1 : 0;
    a.spill_rec = f->spill_rec;
    a.fuse_weights = can_fuse(f) ? 1 : 0;
    if (workgroups_per_cu_out) *workgroups_per_cu_out = update_workgroups_per_cu(a, f->lds_bytes, f->three_per_cu, f->any_layout);
    if (lds_bytes_out) *lds_bytes_out = f->lds_bytes;
    return PHD_OK;
}

extern "C" int phd_update_dev(phd_filter* f, const phd_measurement* d_z, int n_meas)
{
    CHECK_F(f);
    if (n_meas <= 0) return PHD_OK; // the reference skips the update when Z is empty (src/main.cpp:1260)
    int M = std::min(n_meas, f->M_limit); // reference clamps to 256 (src/phdfilter.cu:3390-3394)
    if (can_fuse(f)) {
        FusedWeights fw = {WM_ACCUMULATE | WM_NORMALIZE, 0.0};
        return do_update_merge(f, d_z, M, nullptr, &fw);
    }
    int rc = do_update_merge(f, d_z, M);
    if (rc) return rc;
    return do_weights(f, WM_ACCUMULATE | WM_NORMALIZE, nullptr, 1);
}

extern "C" int phd_update(phd_filter* f, const phd_measurement* z, int n_meas)
{
    CHECK_F(f);
    if (n_meas <= 0) return PHD_OK;
    if (!z) return fail(PHD_ERR_INVALID_ARG, "phd_update: null measurements");
    int M = std::min(n_meas, f->M_limit);
    HIPCHK(hipMemcpyAsync(f->d_z, z, M * sizeof(phd_measurement), hipMemcpyHostToDevice, f->stream));
    return phd_update_dev(f, f->d_z, M);
}

// phdPredict + phdUpdateSynth of one step in ONE launch (the vehicle predict fused in front of the update kernel, the weight
// normalisation as its tail): the same results as phd_predict_ackerman followed by phd_update — the staged calls are what it
// falls back to where the fused launch does not apply (no scan, the particle shotgun, filters that cannot fuse).
extern "C" int phd_predict_update(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* noise, const phd_measurement* z, int n_meas)
{
    CHECK_F(f);
    if (n_meas > 0 && !z) return fail(PHD_ERR_INVALID_ARG, "phd_predict_update: null measurements");
    if (n_meas <= 0 || f->cfg.nPredictParticles > 1 || !can_fuse(f)) {
        const int rc = phd_predict_ackerman(f, u, noise);
        if (rc) return rc;
        return phd_update(f, z, n_meas);
    }
    const int M = std::min(n_meas, f->M_limit);
    if (noise) HIPCHK(hipMemcpyAsync(f->d_noise, noise, (size_t)f->n * sizeof(phd_ackerman_noise), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_z, z, M * sizeof(phd_measurement), hipMemcpyHostToDevice, f->stream));
    FusedPredict fp = {u, noise ? f->d_noise : nullptr};
    FusedWeights fw = {WM_ACCUMULATE | WM_NORMALIZE, 0.0};
    return do_update_merge(f, f->d_z, M, &fp, &fw);
}

extern "C" int phd_neff(phd_filter* f, float* neff_out)
{
    CHECK_F(f);
    if (!neff_out) return fail(PHD_ERR_INVALID_ARG, "null output");
    bool fr = f->frozen;
    f->frozen = true; // read-only evaluation
    int rc = do_weights(f, 0, nullptr, 1);
    f->frozen = fr;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(neff_out, f->neff, sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (*neff_out != *neff_out) return fail(PHD_ERR_NAN, "nan weights detected"); // src/main.cpp:1307-1311
    return PHD_OK;
}

extern "C" int phd_resample(phd_filter* f, const double* uniforms, int n_uniforms, int32_t* idx_out)
{
    CHECK_F(f);
    // resampleParticles(particles, config.n_particles) (src/main.cpp:1289): back to the configured count
    const int n_new = (f->n == f->n_base || f->n_global != f->n_base) ? f->n : f->n_base;
    if (!uniforms || (n_uniforms != 1 && n_uniforms != n_new))
        return fail(PHD_ERR_INVALID_ARG, "phd_resample: n_uniforms must be 1 (systematic) or n_particles (stratified)");
    if (n_uniforms != 1)
        HIPCHK(hipMemcpyAsync(f->d_uniforms, uniforms, n_uniforms * sizeof(double), hipMemcpyHostToDevice, f->stream));
    int rc = do_weights(f, WM_RESAMPLE_FORCE | WM_COMMIT, f->d_uniforms, n_uniforms, uniforms[0]);
    if (rc) return rc;
    if (!f->frozen) f->n = n_new;
    if (idx_out) {
        HIPCHK(hipMemcpyAsync(idx_out, f->idx, n_new * sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    return PHD_OK;
}

extern "C" int phd_resample_if_needed(phd_filter* f, double uniform, int had_measurements, int32_t* did_resample_out,
                                      int32_t* idx_out)
{
    CHECK_F(f);
    // trigger of src/main.cpp:1286: (nEff <= resample_threshold and the step had measurements), decided
    // on the device, OR more than 5*n_particles particles (shotgun growth), decided here
    const bool grown = f->n != f->n_base && f->n_global == f->n_base;
    const bool force = grown && f->n > 5 * f->n_base;
    const int n_before = f->n, n_new = grown ? f->n_base : f->n;
    int rc = do_weights(f, (force ? WM_RESAMPLE_FORCE : WM_RESAMPLE_AUTO) | WM_COMMIT | (had_measurements ? WM_HAD_MEAS : 0),
                        nullptr, 1, uniform);
    if (rc) return rc;
    int did = force ? 1 : 0;
    if (did_resample_out || (grown && !force)) { // with a grown particle set the count depends on the decision
        HIPCHK(hipMemcpyAsync(&did, f->did, sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    if (did_resample_out) *did_resample_out = did;
    if (grown && did && !f->frozen) f->n = n_new;
    if (idx_out) {
        HIPCHK(hipMemcpyAsync(idx_out, f->idx, (did ? n_new : n_before) * sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    return PHD_OK;
}

extern "C" int phd_step_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise,
                            const phd_measurement* d_z, int n_meas, double uniform, int force_resample)
{
    CHECK_F(f);
    int rc;
    int M = std::min(n_meas, f->M_limit);
    if (f->cfg.nPredictParticles > 1) {
        // particle shotgun (src/phdfilter.cu:797-823,1185-1238): the fused in-kernel predict is 1:1, so this step is
        // the staged sequence — predict multiplies the set by k (log-weights - log k, maps shared through the parent
        // indirection), the update runs on the grown set, and the resample is run_synth's trigger
        // (src/main.cpp:1286: nEff <= threshold with measurements, or N > 5 n_particles) back to n_particles
        rc = do_predict(f, u, d_noise);
        if (rc) return rc;
        if (M > 0) {
            rc = phd_update_dev(f, d_z, M);
            if (rc) return rc;
        }
        if (force_resample) {
            const int n_new = (f->n == f->n_base || f->n_global != f->n_base) ? f->n : f->n_base;
            rc = do_weights(f, WM_RESAMPLE_FORCE | WM_COMMIT, nullptr, 1, uniform);
            if (rc) return rc;
            if (!f->frozen) f->n = n_new;
            return PHD_OK;
        }
        return phd_resample_if_needed(f, uniform, M > 0, nullptr, nullptr);
    }
    int mode = WM_COMMIT | (force_resample ? WM_RESAMPLE_FORCE : WM_RESAMPLE_AUTO);
    if (M <= 0) {
        rc = do_predict(f, u, d_noise); // no measurements: predict only (src/main.cpp:1244-1260)
        if (rc) return rc;
    } else {
        FusedPredict fp = {u, d_noise};
        mode |= WM_ACCUMULATE | WM_NORMALIZE | WM_HAD_MEAS;
        if (can_fuse(f)) {
            FusedWeights fw = {mode, uniform};
            return do_update_merge(f, d_z, M, &fp, &fw);
        }
        rc = do_update_merge(f, d_z, M, &fp);
        if (rc) return rc;
        mode |= WM_ACCUMULATE | WM_NORMALIZE | WM_HAD_MEAS;
    }
    return do_weights(f, mode, nullptr, 1, uniform); // the uniform travels as a kernel argument
}

extern "C" int phd_expected_pose(phd_filter* f, phd_pose* out)
{
    CHECK_F(f);
    if (!out) return fail(PHD_ERR_INVALID_ARG, "null output");
    HIPCHK(launch_state(f->pose[f->pose_cur], f->logw, f->n, f->state_pose, f->state_argmax, f->stream));
    HIPCHK(hipMemcpyAsync(out, f->state_pose, sizeof(phd_pose), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_map_estimate(phd_filter* f, phd_gaussian2d* out, int capacity, int32_t* n_out, int32_t* particle_out)
{
    CHECK_F(f);
    int am = -1;
    HIPCHK(launch_state(f->pose[f->pose_cur], f->logw, f->n, f->state_pose, f->state_argmax, f->stream));
    HIPCHK(hipMemcpyAsync(&am, f->state_argmax, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (am < 0) return fail(PHD_ERR_NAN, "no finite particle weight");
    if (particle_out) *particle_out = am;
    return phd_get_map(f, am, out, capacity, n_out);
}

// recoverSlamState in one call and ONE host synchronisation (phd_expected_pose + phd_map_estimate + phd_get_particles make
// three): the arg-max particle's map is unpacked by a kernel that takes the index from device memory
extern "C" int phd_state_snapshot(phd_filter* f, phd_pose* expected_out, phd_gaussian2d* map_out, int capacity, int32_t* n_map_out,
                                  int32_t* particle_out, phd_pose* poses_out, float* log_weights_out, phd_step_report* report_out)
{
    CHECK_F(f);
    if (!expected_out || !map_out || !n_map_out) return fail(PHD_ERR_INVALID_ARG, "phd_state_snapshot: null output");
    int rc = ensure_concat(f, (size_t)f->cap);
    if (rc) return rc;
    int am = -1, nm = 0;
    HIPCHK(launch_state(f->pose[f->pose_cur], f->logw, f->n, f->state_pose, f->state_argmax, f->stream));
    HIPCHK(launch_unpack_one(f->maps[f->cur], f->parent[f->pcur], f->counts[f->cur], f->state_argmax, f->d_concat, f->cap,
                             f->d_tmp_int, f->stream));
    HIPCHK(hipMemcpyAsync(expected_out, f->state_pose, sizeof(phd_pose), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(&am, f->state_argmax, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(&nm, f->d_tmp_int, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    const int ncopy = std::min(capacity, f->cap);
    if (ncopy > 0)
        HIPCHK(hipMemcpyAsync(map_out, f->d_concat, (size_t)ncopy * sizeof(phd_gaussian2d), hipMemcpyDeviceToHost, f->stream));
    if (poses_out) HIPCHK(hipMemcpyAsync(poses_out, f->pose[f->pose_cur], f->n * sizeof(phd_pose), hipMemcpyDeviceToHost, f->stream));
    if (log_weights_out) HIPCHK(hipMemcpyAsync(log_weights_out, f->logw, f->n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    unsigned raw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (report_out) HIPCHK(hipMemcpyAsync(raw, f->report, sizeof(raw), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (report_out) {
        // the step's report rides on the same download: capacity overflow, the fused step's time-out, and the nEff / resample
        // decision the step's weights routine left (src/main.cpp:1281-1311: the NaN exit tests the PRE-resample nEff)
        rc = interpret_report(f, raw, report_out);
        if (rc) return rc;
        if (report_out->neff != report_out->neff) return fail(PHD_ERR_NAN, "nan weights detected");
    }
    if (am < 0) return fail(PHD_ERR_NAN, "no finite particle weight");
    *n_map_out = nm;
    if (particle_out) *particle_out = am;
    if (nm > capacity) return fail(PHD_ERR_CAPACITY, "phd_state_snapshot: map larger than the output buffer");
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// The same snapshot WITHOUT the host synchronisation, for a driver loop that keeps the device busy (include/phdslam.h).
// run_synth logs between the update and the resample (src/main.cpp:1271-1297): that ORDER of contents is kept — the block is
// captured on the filter's stream after the update, the report (nEff, the resample decision) is added after the resample launch —
// but the host no longer waits: a second stream downloads the block while the filter's stream runs the next step.
// ---------------------------------------------------------------------------------------------
static size_t snap_words(const phd_filter* f) { return (size_t)PHD_SNAP_HEADER_WORDS + (size_t)7 * f->cap + (size_t)f->cn_len + (size_t)8 * f->n_max; }

static int ensure_snapshot(phd_filter* f)
{
    if (f->copy_stream) return PHD_OK;
    HIPCHK(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipEventCreateWithFlags(&f->snap_ready[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&f->snap_done[k], hipEventDisableTiming));
        HIPCHK(dalloc(&f->snap_dev[k], snap_words(f)));
        HIPCHK(hipHostMalloc((void**)&f->snap_host[k], snap_words(f) * 4, hipHostMallocDefault));
    }
    return PHD_OK;
}

extern "C" void* phd_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(PHD_ERR_HIP, "phd_host_alloc: hipHostMalloc failed"); return nullptr; }
    return p;
}

extern "C" void phd_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int phd_snapshot_capture(phd_filter* f, int slot)
{
    CHECK_F(f);
    if (slot < 0 || slot > 1) return fail(PHD_ERR_INVALID_ARG, "phd_snapshot_capture: slot must be 0 or 1");
    if (f->snap_state[slot] != 0) return fail(PHD_ERR_INVALID_ARG, "phd_snapshot_capture: the slot holds a snapshot that was not waited for");
    int rc = ensure_snapshot(f);
    if (rc) return rc;
    // ONE launch: phd_state_snapshot's two kernels restated operation for operation (the same bits) + the copies, straight
    // into the slot's staging block (phd_snapshot.hip)
    HIPCHK(launch_snapshot(f->pose[f->pose_cur], f->logw, f->n, f->n_max, f->maps[f->cur], f->parent[f->pcur], f->counts[f->cur], f->cap,
                           f->cphd ? f->cn[f->cur] : nullptr, f->cn_len, f->snap_dev[slot], f->stream));
    f->snap_state[slot] = 1;
    f->snap_n[slot] = f->n;
    return PHD_OK;
}

extern "C" int phd_snapshot_send(phd_filter* f, int slot, int want_resample_idx)
{
    CHECK_F(f);
    if (slot < 0 || slot > 1 || f->snap_state[slot] != 1) return fail(PHD_ERR_INVALID_ARG, "phd_snapshot_send: no captured snapshot in this slot");
    // the report as it stands NOW on the filter's stream (after the resample launch: nEff of the pre-resample weights and the
    // decision), and — for the 7-line log — the parent indices of the resample
    HIPCHK(hipMemcpyAsync(f->snap_dev[slot] + 16, f->report, 8 * 4, hipMemcpyDeviceToDevice, f->stream));
    size_t words = (size_t)PHD_SNAP_HEADER_WORDS + (size_t)7 * f->cap + (size_t)f->cn_len + (size_t)7 * f->n_max;
    f->snap_idx[slot] = want_resample_idx != 0;
    if (want_resample_idx) {
        HIPCHK(hipMemcpyAsync(f->snap_dev[slot] + words, f->idx, (size_t)f->n * 4, hipMemcpyDeviceToDevice, f->stream));
        words += (size_t)f->n;
    }
    HIPCHK(hipEventRecord(f->snap_ready[slot], f->stream));
    HIPCHK(hipStreamWaitEvent(f->copy_stream, f->snap_ready[slot], 0));
    HIPCHK(hipMemcpyAsync(f->snap_host[slot], f->snap_dev[slot], words * 4, hipMemcpyDeviceToHost, f->copy_stream));
    HIPCHK(hipEventRecord(f->snap_done[slot], f->copy_stream));
    f->snap_state[slot] = 2;
    return PHD_OK;
}

extern "C" int phd_snapshot_wait(phd_filter* f, int slot, phd_snapshot_view* out)
{
    CHECK_F(f);
    if (!out) return fail(PHD_ERR_INVALID_ARG, "phd_snapshot_wait: null output");
    if (slot < 0 || slot > 1 || f->snap_state[slot] != 2) return fail(PHD_ERR_INVALID_ARG, "phd_snapshot_wait: nothing was sent from this slot");
    HIPCHK(hipEventSynchronize(f->snap_done[slot]));
    f->snap_state[slot] = 0;
    const float* h = f->snap_host[slot];
    memset(out, 0, sizeof(*out));
    out->expected = (const phd_pose*)h;
    out->particle = ((const int32_t*)h)[6];
    out->n_map = ((const int32_t*)h)[7];
    out->n_particles = ((const int32_t*)h)[8];
    out->map = (const phd_gaussian2d*)(h + PHD_SNAP_HEADER_WORDS);
    out->cardinality = f->cphd ? h + PHD_SNAP_HEADER_WORDS + (size_t)7 * f->cap : nullptr;
    out->cardinality_len = f->cphd ? f->cn_len : 0;
    out->poses = (const phd_pose*)(h + PHD_SNAP_HEADER_WORDS + (size_t)7 * f->cap + (size_t)f->cn_len);
    out->log_weights = h + PHD_SNAP_HEADER_WORDS + (size_t)7 * f->cap + (size_t)f->cn_len + (size_t)6 * f->n_max;
    out->resample_idx = f->snap_idx[slot] ? (const int32_t*)(out->log_weights + f->n_max) : nullptr;
    unsigned raw[8];
    memcpy(raw, h + 16, sizeof(raw));
    int rc = interpret_report(f, raw, &out->report);
    if (rc) return rc;
    if (out->report.neff != out->report.neff) return fail(PHD_ERR_NAN, "nan weights detected");   // src/main.cpp:1307-1311
    if (out->particle < 0) return fail(PHD_ERR_NAN, "no finite particle weight");
    if (out->n_map > f->cap) return fail(PHD_ERR_CAPACITY, "phd_snapshot_wait: map larger than the map capacity");
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// expected-a-posteriori map (config.mapEstimate & 2): computeExpectedMap, src/main.cpp:290-316,
// with reduceGaussianMixture (src/gm_reduce.cpp:57-134) run on the device (phd_eap.hip)
// ---------------------------------------------------------------------------------------------
static int gm_fetch(phd_filter* f, const phd_gaussian2d* d_res, int K, phd_gaussian2d* out, int capacity, int32_t* n_out)
{
    if (n_out) *n_out = K;
    if (K > capacity) return fail(PHD_ERR_CAPACITY, "expected map: output buffer too small (n_out holds the size needed)");
    if (K > 0 && out) {
        HIPCHK(hipMemcpyAsync(out, d_res, (size_t)K * sizeof(phd_gaussian2d), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    return PHD_OK;
}

extern "C" int phd_expected_map_concat_dev(phd_filter* f, float** d_planes, int64_t* total_out)
{
    CHECK_F(f);
    if (!d_planes || !total_out) return fail(PHD_ERR_INVALID_ARG, "phd_expected_map_concat_dev: null output");
    if (!f->gm) f->gm = gm_workspace_create();
    std::vector<int32_t> sizes(f->n);
    int rc = phd_get_map_sizes(f, sizes.data());
    if (rc) return rc;
    std::vector<int> off(f->n + 1, 0);
    for (int p = 0; p < f->n; ++p) off[p + 1] = off[p] + sizes[p];
    const size_t T = (size_t)off[f->n];
    *total_out = (int64_t)T;
    *d_planes = nullptr;
    if (T == 0) return PHD_OK;
    float* buf = gm_concat_buffer(f->gm, T, f->stream);
    int* d_off = gm_offsets_buffer(f->gm, (size_t)f->n_max + 1, f->stream);
    if (!buf || !d_off) return fail(PHD_ERR_HIP, "expected map: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_off, off.data(), (f->n + 1) * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(launch_eap_concat(f->maps[f->cur], f->counts[f->cur], f->parent[f->pcur], f->logw, d_off, f->cap, f->n, buf, T,
                             f->stream));
    HIPCHK(hipStreamSynchronize(f->stream)); // `off` is pageable host memory
    *d_planes = buf;
    return PHD_OK;
}

extern "C" int phd_gm_reduce_dev(phd_filter* f, const float* d_planes, int64_t total, int n_planes, float min_distance,
                                 phd_gaussian2d* out, int capacity, int32_t* n_out)
{
    CHECK_F(f);
    if (total < 0 || (n_planes != 6 && n_planes != 7) || (total > 0 && !d_planes) || capacity < 0)
        return fail(PHD_ERR_INVALID_ARG, "phd_gm_reduce_dev: bad argument");
    if (!f->gm) f->gm = gm_workspace_create();
    const size_t T = (size_t)total;
    const float* in[7] = {d_planes, d_planes + T, d_planes + 2 * T, d_planes + 3 * T, d_planes + 4 * T,
                          n_planes == 7 ? d_planes + 5 * T : d_planes + 4 * T, d_planes + (size_t)(n_planes - 1) * T};
    int K = 0;
    const phd_gaussian2d* d_res = nullptr;
    HIPCHK(gm_reduce_device(f->gm, in, T, min_distance, f->stream, &K, &f->gm_rounds, &d_res));
    return gm_fetch(f, d_res, K, out, capacity, n_out);
}

extern "C" int phd_gm_reduce(phd_filter* f, const phd_gaussian2d* in, int64_t n, float min_distance, phd_gaussian2d* out,
                             int capacity, int32_t* n_out)
{
    CHECK_F(f);
    if (n < 0 || (n > 0 && !in)) return fail(PHD_ERR_INVALID_ARG, "phd_gm_reduce: bad argument");
    if (n == 0) { if (n_out) *n_out = 0; return PHD_OK; }
    const size_t T = (size_t)n;
    std::vector<float> planes(7 * T);
    for (size_t i = 0; i < T; ++i) {
        planes[i] = in[i].weight;
        planes[T + i] = in[i].mean[0];
        planes[2 * T + i] = in[i].mean[1];
        planes[3 * T + i] = in[i].cov[0];
        planes[4 * T + i] = in[i].cov[1];
        planes[5 * T + i] = in[i].cov[2];
        planes[6 * T + i] = in[i].cov[3];
    }
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, planes.size() * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d, planes.data(), planes.size() * sizeof(float), hipMemcpyHostToDevice, f->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
    int rc = e == hipSuccess ? phd_gm_reduce_dev(f, d, n, 7, min_distance, out, capacity, n_out) : fail(PHD_ERR_HIP, hipGetErrorString(e));
    (void)hipStreamSynchronize(f->stream);
    (void)hipFree(d);
    return rc;
}

extern "C" int phd_expected_map(phd_filter* f, phd_gaussian2d* out, int capacity, int32_t* n_out)
{
    CHECK_F(f);
    float* d = nullptr;
    int64_t T = 0;
    int rc = phd_expected_map_concat_dev(f, &d, &T);
    if (rc) return rc;
    if (T == 0) { if (n_out) *n_out = 0; return PHD_OK; }                  // "no features", src/main.cpp:308-313
    return phd_gm_reduce_dev(f, d, T, 6, f->cfg.minSeparation, out, capacity, n_out);
}

extern "C" int phd_debug_gm_rounds(phd_filter* f) { return f ? f->gm_rounds : 0; }
extern "C" int phd_debug_copy_free_resamples(phd_filter* f) { return f ? f->copy_free_resamples : 0; }
extern "C" int phd_debug_update_instantiation(phd_filter* f) { return f ? f->last_update_fn : -1; }

// ---------------------------------------------------------------------------------------------
// CPHD variant: the per-particle cardinality distributions (SynthSLAM::cardinalities, src/slamtypes.h:296)
// ---------------------------------------------------------------------------------------------
extern "C" int phd_cardinality_length(const phd_filter* f) { return f ? f->cn_len : 0; }

extern "C" int phd_get_cardinalities(phd_filter* f, float* out)
{
    CHECK_F(f);
    if (!f->cphd) return fail(PHD_ERR_UNSUPPORTED, "phd_get_cardinalities: filter_type is not CPHD");
    if (!out) return fail(PHD_ERR_INVALID_ARG, "null output");
    int rc = materialize_parents(f);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, f->cn[f->cur], (size_t)f->n * f->cn_len * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_set_cardinalities(phd_filter* f, const float* in)
{
    CHECK_F(f);
    if (!f->cphd) return fail(PHD_ERR_UNSUPPORTED, "phd_set_cardinalities: filter_type is not CPHD");
    if (!in) return fail(PHD_ERR_INVALID_ARG, "null input");
    int rc = materialize_parents(f);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(f->cn[f->cur], in, (size_t)f->n * f->cn_len * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

// cn_estimate of recoverSlamState (src/main.cpp:360): the cardinality distribution of the arg-max-weight particle
extern "C" int phd_cardinality_estimate(phd_filter* f, float* out, int32_t* particle_out)
{
    CHECK_F(f);
    if (!f->cphd) return fail(PHD_ERR_UNSUPPORTED, "phd_cardinality_estimate: filter_type is not CPHD");
    if (!out) return fail(PHD_ERR_INVALID_ARG, "null output");
    int am = -1, src = 0;
    HIPCHK(launch_state(f->pose[f->pose_cur], f->logw, f->n, f->state_pose, f->state_argmax, f->stream));
    HIPCHK(hipMemcpyAsync(&am, f->state_argmax, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (am < 0) return fail(PHD_ERR_NAN, "no finite particle weight");
    HIPCHK(hipMemcpyAsync(&src, f->parent[f->pcur] + am, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    HIPCHK(hipMemcpyAsync(out, f->cn[f->cur] + (size_t)src * f->cn_len, f->cn_len * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (particle_out) *particle_out = am;
    return PHD_OK;
}

extern "C" int phd_logweights_dev(phd_filter* f, float** d) { CHECK_F(f); if (!d) return fail(PHD_ERR_INVALID_ARG, "null"); *d = f->logw; return PHD_OK; }
// The un-normalised weights of a shard's local step written straight into the caller's buffer (n_max floats; NULL: the filter's own
// again): with the shard's segment of the all-gather's receive buffer as the target the collective is IN PLACE — on one rank it
// disappears, on N ranks a 1/N-th of its bytes and the self copy do.
extern "C" int phd_set_raw_target(phd_filter* f, float* d_raw)
{
    CHECK_F(f);
    f->logw_raw = d_raw ? d_raw : f->logw_raw_own;
    return PHD_OK;
}
extern "C" int phd_raw_logweights_dev(phd_filter* f, float** d) { CHECK_F(f); if (!d) return fail(PHD_ERR_INVALID_ARG, "null"); *d = f->logw_raw; return PHD_OK; }

extern "C" int phd_set_frozen(phd_filter* f, int freeze)
{
    CHECK_F(f);
    f->frozen = freeze != 0;
    f->pose_for_update = nullptr;
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: global normalise / resample over the all-gathered weight vector
// ---------------------------------------------------------------------------------------------
// Per step on every rank:  predict -> phd_update_local_dev (update+merge, raw = logw + dlogw)
//   -> all_gather(raw) [RCCL, by the caller] -> phd_global_normalize -> (if nEff says so)
//   phd_global_resample_indices -> export / all_to_all / apply_parents / import -> phd_finish_resample
extern "C" int phd_update_local_dev(phd_filter* f, const phd_measurement* d_z, int n_meas)
{
    CHECK_F(f);
    if (n_meas <= 0) {
        HIPCHK(hipMemcpyAsync(f->logw_raw, f->logw, f->n * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
        return PHD_OK;
    }
    int M = std::min(n_meas, f->M_limit);
    int rc = do_update_merge(f, d_z, M);
    if (rc) return rc;
    return do_weights(f, WM_ACCUMULATE, nullptr, 1); // raw_out = logw + dlogw, no normalisation
}

// predict + update + prune + merge of the local shard in ONE launch, raw = logw + dlogw written by the same kernel
extern "C" int phd_step_local_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise,
                                  const phd_measurement* d_z, int n_meas)
{
    CHECK_F(f);
    int M = std::min(n_meas, f->M_limit);
    if (M <= 0) {
        int rc = do_predict(f, u, d_noise);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(f->logw_raw, f->logw, f->n * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
        return PHD_OK;
    }
    if (f->n != f->n_base || f->cfg.nPredictParticles > 1)
        return fail(PHD_ERR_UNSUPPORTED, "phd_step_local_dev: particle shotgun shards (n_predict_particles > 1) use the staged calls");
    FusedPredict fp = {u, d_noise};
    f->want_raw = true;
    int rc = do_update_merge(f, d_z, M, &fp);
    f->want_raw = false;
    return rc;
}

extern "C" int phd_global_normalize(phd_filter* f, const float* d_all_logw, int n_global, float* neff_out)
{
    CHECK_F(f);
    if (n_global != ng_cur(f)) return fail(PHD_ERR_INVALID_ARG, "phd_global_normalize: n_global mismatch");
    WeightArgs w;
    memset(&w, 0, sizeof(w));
    w.logw_in = d_all_logw;
    w.logw = f->logw_scratch; // [n_global] normalised copy (identical on every rank)
    w.n = n_global;
    w.n_new = 0;
    w.mode = WM_NORMALIZE;
    w.idx_out = f->idx;
    w.neff_out = f->neff;
    w.did_resample = f->did;
    w.n_weight_norm = f->n_global;
    t_begin(f, PHD_K_WEIGHTS);
    HIPCHK(launch_weights_f(f, w));
    t_end(f);
    if (!f->frozen)
        HIPCHK(hipMemcpyAsync(f->logw, f->logw_scratch + off_cur(f), f->n * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
    if (neff_out) {
        HIPCHK(hipMemcpyAsync(neff_out, f->neff, sizeof(float), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        if (*neff_out != *neff_out) return fail(PHD_ERR_NAN, "nan weights detected");
    }
    return PHD_OK;
}

extern "C" int phd_global_resample_indices(phd_filter* f, const float* d_all_logw_normalized, int n_global,
                                           const double* uniforms, int n_uniforms, int32_t* idx_out)
{
    CHECK_F(f);
    if (n_global != f->n_global) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_indices: n_global mismatch");
    if (!uniforms || (n_uniforms != 1 && n_uniforms != n_global)) return fail(PHD_ERR_INVALID_ARG, "bad uniforms");
    const float* src = d_all_logw_normalized ? d_all_logw_normalized : f->logw_scratch;
    if (n_uniforms != 1)
        HIPCHK(hipMemcpyAsync(f->d_uniforms, uniforms, n_uniforms * sizeof(double), hipMemcpyHostToDevice, f->stream));
    WeightArgs w;
    memset(&w, 0, sizeof(w));
    w.u0 = uniforms[0];
    w.logw_in = src;
    w.logw = const_cast<float*>(src);
    w.n = n_global;
    w.n_new = n_global;
    w.mode = WM_RESAMPLE_FORCE;
    w.uniforms = f->d_uniforms;
    w.n_uniforms = n_uniforms;
    w.cdf = f->cdf;
    w.idx_out = f->idx;
    w.neff_out = f->neff;
    w.did_resample = f->did;
    w.n_weight_norm = f->n_global;
    t_begin(f, PHD_K_WEIGHTS);
    HIPCHK(launch_weights_f(f, w));
    t_end(f);
    if (idx_out) {
        HIPCHK(hipMemcpyAsync(idx_out, f->idx, n_global * sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    return PHD_OK;
}

extern "C" size_t phd_particle_pack_bytes(const phd_filter* f) { return f ? (size_t)(8 + 6 * f->cap + f->cn_len) * 4 : 0; }

extern "C" int phd_export_particles_dev(phd_filter* f, const int32_t* particles, int n, void* d_buffer)
{
    CHECK_F(f);
    if (n <= 0) return PHD_OK;
    // one particle may be the parent of slots on every other rank: up to n_global exports
    if (n > std::max(f->n, f->n_global)) return fail(PHD_ERR_INVALID_ARG, "phd_export_particles_dev: n > n_global");
    HIPCHK(hipMemcpyAsync(f->d_sizes, particles, n * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(launch_export(f->maps[f->cur], f->counts[f->cur], f->parent[f->pcur],
                         f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur], f->d_sizes,
                         d_buffer, f->cap, phd_particle_pack_bytes(f), n, f->stream));
    if (f->cphd) // the cardinality row travels behind the slab
        HIPCHK(launch_copy_rows(f->cn[f->cur], f->cn_len, f->d_sizes, f->parent[f->pcur], (float*)d_buffer + 8 + 6 * f->cap,
                                phd_particle_pack_bytes(f) / 4, nullptr, f->cn_len, n, f->stream));
    return PHD_OK;
}

// local part of copy_particles after a global resample: next buffers <- parents that live on this rank
extern "C" int phd_apply_parents(phd_filter* f, const int32_t* local_parent)
{
    CHECK_F(f);
    HIPCHK(hipMemcpyAsync(f->d_tmp_int, local_parent, f->n * sizeof(int), hipMemcpyHostToDevice, f->stream));
    const int pnext = (f->pose_cur + (f->pose_for_update ? 2 : 1)) % 3; // frozen fused predict parks the predicted poses in +1
    HIPCHK(launch_gather_maps(f->maps[f->cur], f->counts[f->cur], f->parent[f->pcur], f->d_tmp_int, f->maps[f->cur ^ 1],
                              f->counts[f->cur ^ 1], f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur], f->pose[pnext],
                              f->cap, f->n, f->stream));
    if (f->cphd)
        HIPCHK(launch_copy_rows(f->cn[f->cur], f->cn_len, f->d_tmp_int, f->parent[f->pcur], f->cn[f->cur ^ 1], f->cn_len, nullptr,
                                f->cn_len, f->n, f->stream));
    return PHD_OK;
}

extern "C" int phd_import_particles_dev(phd_filter* f, const int32_t* slots, int n, const void* d_buffer)
{
    CHECK_F(f);
    if (n <= 0) return PHD_OK;
    if (n > f->n) return fail(PHD_ERR_INVALID_ARG, "phd_import_particles_dev: n > n_particles");
    const int pnext = (f->pose_cur + (f->pose_for_update ? 2 : 1)) % 3;
    HIPCHK(hipMemcpyAsync(f->d_sizes, slots, n * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(launch_import(f->maps[f->cur ^ 1], f->counts[f->cur ^ 1], f->pose[pnext], f->d_sizes, d_buffer, f->cap,
                         phd_particle_pack_bytes(f), n, f->stream));
    if (f->cphd)
        HIPCHK(launch_copy_rows((const float*)d_buffer + 8 + 6 * f->cap, phd_particle_pack_bytes(f) / 4, nullptr, nullptr,
                                f->cn[f->cur ^ 1], f->cn_len, f->d_sizes, f->cn_len, n, f->stream));
    return PHD_OK;
}

// the same with a row selection: slot slots[k] takes row rows[k] of the buffer (a received parent that fills several slots)
extern "C" int phd_import_particles_sel_dev(phd_filter* f, const int32_t* slots, const int32_t* rows, int n, const void* d_buffer)
{
    CHECK_F(f);
    if (!rows) return phd_import_particles_dev(f, slots, n, d_buffer);
    if (n <= 0) return PHD_OK;
    if (n > f->n) return fail(PHD_ERR_INVALID_ARG, "phd_import_particles_sel_dev: n > n_particles");
    const int pnext = (f->pose_cur + (f->pose_for_update ? 2 : 1)) % 3;
    HIPCHK(hipMemcpyAsync(f->d_sizes, slots, n * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_tmp_int, rows, n * sizeof(int), hipMemcpyHostToDevice, f->stream)); // after phd_apply_parents' kernel (stream order)
    HIPCHK(launch_import(f->maps[f->cur ^ 1], f->counts[f->cur ^ 1], f->pose[pnext], f->d_sizes, d_buffer, f->cap,
                         phd_particle_pack_bytes(f), n, f->stream, f->d_tmp_int));
    if (f->cphd)
        HIPCHK(launch_copy_rows((const float*)d_buffer + 8 + 6 * f->cap, phd_particle_pack_bytes(f) / 4, f->d_tmp_int, nullptr,
                                f->cn[f->cur ^ 1], f->cn_len, f->d_sizes, f->cn_len, n, f->stream));
    return PHD_OK;
}

extern "C" int phd_finish_resample(phd_filter* f)
{
    CHECK_F(f);
    if (f->frozen) return PHD_OK; // bench protocol: the exchange ran, the snapshot stays
    f->cur ^= 1;
    f->pose_cur = (f->pose_cur + 1) % 3;
    HIPCHK(launch_iota(f->parent[f->pcur], f->n, f->stream));
    f->parent_dirty = false;
    HIPCHK(launch_fill(f->logw, (float)(-log((double)f->n_global)), f->n, f->stream)); // src/slamtypes.h:327
    return PHD_OK; // stream-ordered: no host synchronisation
}

static int ensure_send_buffer(phd_filter* f, size_t need)
{
    if (need > f->send_buf_bytes) {
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->send_buf) hipFree(f->send_buf);
        f->send_buf = nullptr;
        f->send_buf_bytes = 0;
        HIPCHK(hipMalloc(&f->send_buf, need * 2));
        f->send_buf_bytes = need * 2;
    }
    return PHD_OK;
}

// The exchange above in two calls (what the multi-GPU host runs per resampling step):
//   begin: indices on the device (identical on every rank) -> download -> this rank's part of the plan
//          (a pure function of the indices: slot j of rank q receives particle idx[q n + j], owned by rank
//          idx / n) -> export of the particles other ranks need into one send buffer, grouped by
//          destination rank in the order the destination's slots appear
//   [caller: all_to_all_single(recv, send, recv_counts, send_counts) over RCCL]
//   end:   local parents gathered, received particles imported into their slots, weights <- -log N
// phd_global_resample_begin in two halves, for a host that drives SEVERAL shards (phd_multi.cpp): every shard enqueues
// the launch; the indices — identical on every shard — are downloaded ONCE and every shard plans from the same host copy.
//   launch: normalise the gathered raw weights (d_all_raw_logw != NULL) and draw the global indices on the device, no sync
//   plan:   this shard's part of the migration from host indices + export of what other shards need
static int global_resample_launch(phd_filter* f, const float* d_all_logw, bool normalize, double uniform, int32_t** d_idx_out)
{
    const int ng = ng_cur(f);                     // (a grown set — particle shotgun — is resampled back to global_particles)
    WeightArgs w;
    memset(&w, 0, sizeof(w));
    w.u0 = uniform;
    w.logw_in = d_all_logw ? d_all_logw : f->logw_scratch;
    w.logw = f->logw_scratch;
    w.n = ng;
    w.n_new = f->n_global;
    w.mode = (normalize ? WM_NORMALIZE : 0) | WM_RESAMPLE_FORCE;
    w.uniforms = f->d_uniforms;
    w.n_uniforms = 1;
    w.cdf = f->cdf;
    w.idx_out = f->idx;
    w.neff_out = f->neff;
    w.did_resample = f->did;
    w.n_weight_norm = f->n_global;
    t_begin(f, PHD_K_WEIGHTS);
    HIPCHK(launch_weights_f(f, w));
    t_end(f);
    if (d_idx_out) *d_idx_out = f->idx;
    return PHD_OK;
}

extern "C" int phd_global_resample_launch(phd_filter* f, const float* d_all_raw_logw, double uniform, int32_t** d_idx_out)
{
    CHECK_F(f);
    return global_resample_launch(f, d_all_raw_logw, d_all_raw_logw != nullptr, uniform, d_idx_out);
}

// the gathered vector is the shards' CURRENT normalised log-weights (a resample that no update precedes: nothing to
// normalise — a second normalisation would move the last bits and with them, now and then, an index)
extern "C" int phd_global_resample_launch_normalized(phd_filter* f, const float* d_all_logw, double uniform, int32_t** d_idx_out)
{
    CHECK_F(f);
    if (!d_all_logw) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_launch_normalized: null weights");
    return global_resample_launch(f, d_all_logw, false, uniform, d_idx_out);
}

extern "C" int phd_global_resample_plan(phd_filter* f, const int32_t* idx, int world, int rank, int32_t* send_counts,
                                        int32_t* recv_counts, void** d_send_buffer)
{
    CHECK_F(f);
    if (world < 1 || rank < 0 || rank >= world || f->n_global != f->n * world || f->global_offset != rank * f->n)
        return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_plan: world/rank do not match the filter's shard");
    if (!idx || !send_counts || !recv_counts || !d_send_buffer) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_plan: null argument");
    if (f->n != f->n_base) return fail(PHD_ERR_UNSUPPORTED, "phd_global_resample_plan: a grown particle set (n_predict_particles > 1) migrates by phd_global_resample_pull");
    const int n = f->n, off = rank * n;
    f->plan_local_parent.assign(n, -1);
    f->plan_send.clear();
    f->plan_recv_slots.clear();
    for (int j = 0; j < n; ++j)
        if ((unsigned)(idx[off + j] - off) < (unsigned)n) f->plan_local_parent[j] = idx[off + j] - off;
    // A parent travels ONCE per destination rank, however many of that rank's slots it fills (resampling is called when the
    // weights have degenerated: a few parents fill most slots): both sides walk the destination's slots in order and skip
    // a parent equal to the previous one from the same source (systematic indices are non-decreasing, so that is every
    // repeat; were they not, the rule is still the same on both sides).  The receiver fans a row out to its slots.
    f->plan_recv_rows.clear();
    std::vector<int32_t>& first_slot = f->plan_row_first;     // the first slot each received row fills (the copy-free form parks it there)
    first_slot.clear();
    int row = -1;
    for (int r = 0; r < world; ++r) {
        send_counts[r] = recv_counts[r] = 0;
        if (r == rank) continue;
        int last = -1;
        for (int g = r * n; g < (r + 1) * n; ++g)                  // what rank r needs from this rank, in r's slot order
            if ((unsigned)(idx[g] - off) < (unsigned)n && idx[g] != last) { f->plan_send.push_back(idx[g] - off); ++send_counts[r]; last = idx[g]; }
        last = -1;
        for (int j = 0; j < n; ++j)                                 // what this rank needs from rank r, in slot order
            if ((unsigned)(idx[off + j] - r * n) < (unsigned)n) {
                if (idx[off + j] != last) { ++recv_counts[r]; ++row; last = idx[off + j]; first_slot.push_back(j); }
                f->plan_recv_slots.push_back(j);
                f->plan_recv_rows.push_back(row);
            }
    }
    // per-slot form of the same plan for phd_global_resample_end's single launch (pinned: the upload does not stage).  The
    // previous step's upload has completed: the caller synchronised to read this step's indices.
    for (int j = 0; j < n; ++j) { f->h_plan[j] = f->plan_local_parent[j]; f->h_plan[n + j] = -1; f->h_plan[2 * n + j] = -1; }
    for (size_t k = 0; k < f->plan_recv_slots.size(); ++k) {
        f->h_plan[n + f->plan_recv_slots[k]] = f->plan_recv_rows[k];
        f->h_plan[2 * n + f->plan_recv_slots[k]] = first_slot[f->plan_recv_rows[k]];
    }
    int rc = ensure_send_buffer(f, std::max<size_t>(f->plan_send.size(), 1) * phd_particle_pack_bytes(f));
    if (rc) return rc;
    if (!f->plan_send.empty()) {
        rc = phd_export_particles_dev(f, f->plan_send.data(), (int)f->plan_send.size(), f->send_buf);
        if (rc) return rc;
    }
    *d_send_buffer = f->send_buf;
    return PHD_OK;
}

extern "C" int phd_global_resample_begin(phd_filter* f, const float* d_all_raw_logw, double uniform, int world, int rank,
                                         int32_t* send_counts, int32_t* recv_counts, void** d_send_buffer, int32_t* idx_out)
{
    CHECK_F(f);
    if (world < 1 || rank < 0 || rank >= world || f->n_global != f->n * world || f->global_offset != rank * f->n)
        return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_begin: world/rank do not match the filter's shard");
    if (!send_counts || !recv_counts || !d_send_buffer) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_begin: null output");
    const int ng = f->n_global;
    f->plan_idx.resize(ng);
    int rc;
    if (d_all_raw_logw) {
        // gathered un-normalised weights: normalise and draw the indices in one launch (forced resample)
        rc = phd_global_resample_launch(f, d_all_raw_logw, uniform, nullptr);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(f->plan_idx.data(), f->idx, (size_t)ng * sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    } else {
        rc = phd_global_resample_indices(f, nullptr, ng, &uniform, 1, f->plan_idx.data());
        if (rc) return rc;
    }
    rc = phd_global_resample_plan(f, f->plan_idx.data(), world, rank, send_counts, recv_counts, d_send_buffer);
    if (rc) return rc;
    if (idx_out) memcpy(idx_out, f->plan_idx.data(), (size_t)ng * sizeof(int32_t));
    return PHD_OK;
}

extern "C" int phd_global_resample_end(phd_filter* f, const void* d_recv_buffer)
{
    CHECK_F(f);
    if ((int)f->plan_local_parent.size() != f->n) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_end: no plan (call _begin first)");
    if (!f->plan_recv_slots.empty() && !d_recv_buffer) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_end: null receive buffer");
    // one upload, one launch: local parents, received rows, weights <- -log N, identity indirection (copy_particles,
    // src/slamtypes.h:313-333; the staged trio phd_apply_parents / phd_import_particles_dev / phd_finish_resample does the same
    // in three uploads and four launches)
    const int n = f->n;
    HIPCHK(hipMemcpyAsync(f->d_plan, f->h_plan, 3 * (size_t)n * sizeof(int), hipMemcpyHostToDevice, f->stream));
    const int pnext = (f->pose_cur + (f->pose_for_update ? 2 : 1)) % 3; // a frozen fused predict parks the predicted poses in +1
    if (copy_free_ok(f, !f->plan_recv_slots.empty())) {
        // the copy-free form: local parents by indirection, received rows into the guest slabs; the buffers do not flip
        HIPCHK(launch_resample_end_free(f->parent[f->pcur], f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur], f->d_plan, n,
                                        d_recv_buffer, phd_particle_pack_bytes(f), f->maps_g, f->counts_g, f->cphd ? f->cn_g : nullptr,
                                        guest_offset(f), f->pose[pnext], f->cap, f->frozen ? nullptr : f->logw,
                                        (float)(-log((double)f->n_global)), f->frozen ? f->parent[2] : f->parent[f->pcur ^ 1], f->cn_len,
                                        f->stream));
        f->plan_local_parent.clear();
        f->copy_free_resamples++;
        if (f->frozen) return PHD_OK;
        f->pose_cur = (f->pose_cur + 1) % 3;
        f->pcur ^= 1;
        f->parent_dirty = true;       // the indirection names slabs of the current buffer and guests until the next update
        return PHD_OK;
    }
    HIPCHK(launch_resample_end(f->maps[f->cur], f->counts[f->cur], f->parent[f->pcur],
                               f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur], f->d_plan, n, d_recv_buffer,
                               phd_particle_pack_bytes(f), f->maps[f->cur ^ 1], f->counts[f->cur ^ 1], f->pose[pnext], f->cap,
                               f->frozen ? nullptr : f->logw, (float)(-log((double)f->n_global)),
                               f->frozen ? nullptr : f->parent[f->pcur ^ 1], f->cphd ? f->cn[f->cur] : nullptr,
                               f->cphd ? f->cn[f->cur ^ 1] : nullptr, f->cn_len, f->stream));
    f->plan_local_parent.clear();
    if (f->frozen) return PHD_OK; // bench protocol: the exchange ran, the snapshot stays
    f->cur ^= 1;
    f->pose_cur = (f->pose_cur + 1) % 3;
    f->pcur ^= 1;
    f->parent_dirty = false;
    return PHD_OK; // stream-ordered: no host synchronisation
}

// copy_particles by direct reads of the owners' memory: see include/phdslam.h
extern "C" int phd_peer_view_get(phd_filter* f, phd_peer_view* out)
{
    CHECK_F(f);
    if (!out) return fail(PHD_ERR_INVALID_ARG, "phd_peer_view_get: null output");
    out->maps = f->maps[f->cur];
    out->counts = f->counts[f->cur];
    out->parent = f->parent[f->pcur];
    out->poses = f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur];
    out->cn = f->cphd ? f->cn[f->cur] : nullptr;
    return PHD_OK;
}

static int resample_pull(phd_filter* f, const phd_peer_view* views, int world, int rank, bool by_device_decision);
extern "C" int phd_global_resample_pull(phd_filter* f, const phd_peer_view* views, int world, int rank)
{
    return resample_pull(f, views, world, rank, false);
}

// The nEff-triggered step of a shard with NO host round trip: phd_global_resample_launch_auto normalises the gathered raw
// weights, takes nEff and the reference's decision (src/main.cpp:1286) ON THE DEVICE and leaves either the resampling indices
// or the identity; phd_global_resample_pull_auto is enqueued whatever the decision was — on "no resample" every slot keeps its
// own particle (nothing is copied: the copy-free form) and the normalised weights stay.  The host reads the decision from the
// step report if and when it wants it.
extern "C" int phd_global_resample_auto_supported(phd_filter* f, int world)
{
    return (f && copy_free_ok(f, world > 1)) ? 1 : 0;
}

extern "C" int phd_global_resample_launch_auto(phd_filter* f, const float* d_all_raw_logw, double uniform)
{
    CHECK_F(f);
    if (!d_all_raw_logw) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_launch_auto: null weights");
    if (f->n != f->n_base) return fail(PHD_ERR_UNSUPPORTED, "phd_global_resample_launch_auto: a grown particle set takes the staged calls");
    WeightArgs w;
    memset(&w, 0, sizeof(w));
    w.u0 = uniform;
    w.logw_in = d_all_raw_logw;
    w.logw = f->logw_scratch;
    w.n = ng_cur(f);
    w.n_new = f->n_global;
    w.mode = WM_NORMALIZE | WM_RESAMPLE_AUTO | WM_HAD_MEAS;
    w.resample_thresh = f->cfg.resampleThresh;
    w.uniforms = f->d_uniforms;
    w.n_uniforms = 1;
    w.cdf = f->cdf;
    w.idx_out = f->idx;
    w.neff_out = f->neff;
    w.did_resample = f->did;
    w.n_weight_norm = f->n_global;
    t_begin(f, PHD_K_WEIGHTS);
    HIPCHK(launch_weights_f(f, w));
    t_end(f);
    // (this shard's slice of the normalised vector is adopted by the pull that follows — or replaced by -log N)
    return PHD_OK;
}

extern "C" int phd_global_resample_pull_auto(phd_filter* f, const phd_peer_view* views, int world, int rank)
{
    CHECK_F(f);
    if (!copy_free_ok(f, world > 1))
        return fail(PHD_ERR_UNSUPPORTED, "phd_global_resample_pull_auto: the copy-free form is not available (ask phd_global_resample_auto_supported first)");
    return resample_pull(f, views, world, rank, true);
}

static int resample_pull(phd_filter* f, const phd_peer_view* views, int world, int rank, bool by_device_decision)
{
    CHECK_F(f);
    if (world < 1 || world > PHD_MAX_PEERS || rank < 0 || rank >= world || f->n_global != f->n_base * world || f->global_offset != rank * f->n_base)
        return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_pull: world/rank do not match the filter's shard");
    if (!views) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_pull: null views");
    // n_src: particles every shard holds NOW (grown by the particle shotgun, the same factor everywhere); n_dst: what it holds
    // after the resample (src/main.cpp:1289 resamples back to n_particles)
    const int n_src = f->n, n_dst = f->n_base;
    const int pnext = (f->pose_cur + (f->pose_for_update ? 2 : 1)) % 3; // a frozen fused predict parks the predicted poses in +1
    if (copy_free_ok(f, world > 1)) {
        // the copy-free form (phd_kernels.hip): only the first slot of a remote parent moves a map, into its guest slab
        HIPCHK(launch_resample_pull_free(views, world, f->idx, rank * n_dst, n_src, n_dst, rank, f->maps_g, f->counts_g,
                                         f->cphd ? f->cn_g : nullptr, guest_offset(f), f->pose[pnext], f->cap,
                                         f->frozen ? nullptr : f->logw, (float)(-log((double)f->n_global)),
                                         f->frozen ? f->parent[2] : f->parent[f->pcur ^ 1], f->cn_len, f->stream,
                                         by_device_decision ? f->did : nullptr, by_device_decision ? f->logw_scratch + off_cur(f) : nullptr));
        f->copy_free_resamples++;
        if (f->frozen) return PHD_OK;
        f->n = n_dst;
        f->pose_cur = (f->pose_cur + 1) % 3;
        f->pcur ^= 1;
        f->parent_dirty = true;       // the indirection names slabs of the current buffer and guests until the next update
        return PHD_OK;
    }
    HIPCHK(launch_resample_pull(views, world, f->idx, rank * n_dst, n_src, n_dst, rank, f->maps[f->cur ^ 1], f->counts[f->cur ^ 1],
                                f->pose[pnext], f->cap, f->frozen ? nullptr : f->logw, (float)(-log((double)f->n_global)),
                                f->frozen ? nullptr : f->parent[f->pcur ^ 1], f->cphd ? f->cn[f->cur ^ 1] : nullptr, f->cn_len,
                                f->stream));
    if (f->frozen) return PHD_OK; // bench protocol: the exchange ran, the snapshot stays
    f->n = n_dst;
    f->cur ^= 1;
    f->pose_cur = (f->pose_cur + 1) % 3;
    f->pcur ^= 1;
    f->parent_dirty = false;
    return PHD_OK; // stream-ordered: no host synchronisation
}

// The same exchange for SMALL shards without a host round trip (the "gathered" exchange): every rank all-gathers every
// rank's whole shard — rows of phd_particle_pack_bytes, the un-normalised log-weight in header word 7 — and then
// normalises, draws the (identical) global indices and takes its slots' parents straight out of the gathered rows.
// One fixed-size collective, the indices never leave the device, nothing waits for the host.  The traffic grows with
// world * n * pack bytes per rank, so the host side picks this form only while that is small
// (cuda-phdslam_amd/dist.py); the all-to-all form above moves only the particles that migrate.
static int export_shard(phd_filter* f, void** d_rows, size_t* bytes_out, const float* header_weights);
extern "C" int phd_export_shard_dev(phd_filter* f, void** d_rows, size_t* bytes_out)
{
    CHECK_F(f);
    if (!d_rows) return fail(PHD_ERR_INVALID_ARG, "phd_export_shard_dev: null output");
    return export_shard(f, d_rows, bytes_out, f->logw_raw);
}
// the same with the CURRENT (normalised) log-weights in the row headers: a resample that no update precedes
// (phd_global_resample_gathered with weights_in_rows = 2 uses them as they are)
extern "C" int phd_export_shard_current_dev(phd_filter* f, void** d_rows, size_t* bytes_out)
{
    CHECK_F(f);
    if (!d_rows) return fail(PHD_ERR_INVALID_ARG, "phd_export_shard_current_dev: null output");
    return export_shard(f, d_rows, bytes_out, f->logw);
}
static int export_shard(phd_filter* f, void** d_rows, size_t* bytes_out, const float* header_weights)
{
    const size_t pack = phd_particle_pack_bytes(f), need = (size_t)f->n * pack;
    int rc = ensure_send_buffer(f, need);
    if (rc) return rc;
    HIPCHK(launch_export(f->maps[f->cur], f->counts[f->cur], f->parent[f->pcur],
                         f->pose_for_update ? f->pose_for_update : f->pose[f->pose_cur], nullptr, f->send_buf, f->cap, pack,
                         f->n, f->stream, header_weights));
    if (f->cphd)
        HIPCHK(launch_copy_rows(f->cn[f->cur], f->cn_len, nullptr, f->parent[f->pcur], (float*)f->send_buf + 8 + 6 * f->cap,
                                pack / 4, nullptr, f->cn_len, f->n, f->stream));
    *d_rows = f->send_buf;
    if (bytes_out) *bytes_out = need;
    return PHD_OK;
}

// phd_step_local_dev + phd_export_shard_dev in ONE launch: the update kernel writes each particle's merged map, predicted
// pose, count, raw log-weight (and CPHD cardinality row) straight into its export row.  The updated maps then exist
// ONLY in the rows: the step must be completed by phd_global_resample_gathered (which rebuilds every slot from the
// gathered rows); until then every other call on a live (not frozen) filter is refused.
extern "C" int phd_step_local_rows_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise,
                                       const phd_measurement* d_z, int n_meas, void** d_rows, size_t* bytes_out)
{
    CHECK_F(f);
    if (!d_rows) return fail(PHD_ERR_INVALID_ARG, "phd_step_local_rows_dev: null output");
    const int M = std::min(n_meas, f->M_limit);
    if (f->cfg.nPredictParticles > 1)
        return fail(PHD_ERR_UNSUPPORTED, "phd_step_local_rows_dev: particle shotgun shards (n_predict_particles > 1) use the staged calls");
    if (M <= 0 || f->n != f->n_base) { // no measurements: the staged form
        int rc = phd_step_local_dev(f, u, d_noise, d_z, n_meas);
        if (rc) return rc;
        return phd_export_shard_dev(f, d_rows, bytes_out);
    }
    const size_t need = (size_t)f->n * phd_particle_pack_bytes(f);
    int rc = f->rows_target ? PHD_OK : ensure_send_buffer(f, need);
    if (rc) return rc;
    FusedPredict fp = {u, d_noise};
    f->want_raw = true;
    f->rows_out = true;
    rc = do_update_merge(f, d_z, M, &fp);
    f->want_raw = false;
    f->rows_out = false;
    if (rc) return rc;
    f->rows_pending = !f->frozen;
    *d_rows = f->rows_target ? f->rows_target : f->send_buf;
    if (bytes_out) *bytes_out = need;
    return PHD_OK;
}

extern "C" int phd_set_rows_target(phd_filter* f, void* d_rows)
{
    CHECK_F0(f);
    f->rows_target = d_rows;
    return PHD_OK;
}

// d_all_rows: world * n rows in rank order (the all-gather of every rank's phd_export_shard_dev).
// weights_in_rows != 0: normalise the header weights and draw the indices in one launch (forced resample);
// 0: the indices come from the vector phd_global_normalize left (the nEff-triggered resample of the SLAM loop).
extern "C" int phd_global_resample_gathered(phd_filter* f, const void* d_all_rows, double uniform, int world, int rank,
                                            int weights_in_rows, int32_t* idx_out)
{
    CHECK_F0(f);
    if (world < 1 || rank < 0 || rank >= world || f->n_global != f->n * world || f->global_offset != rank * f->n)
        return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_gathered: world/rank do not match the filter's shard");
    if (!d_all_rows) return fail(PHD_ERR_INVALID_ARG, "phd_global_resample_gathered: null rows");
    if (f->n != f->n_base) return fail(PHD_ERR_UNSUPPORTED, "phd_global_resample_gathered: a grown particle set (n_predict_particles > 1) migrates by phd_global_resample_pull");
    const int ng = f->n_global, n = f->n, off = rank * n;
    const size_t pack = phd_particle_pack_bytes(f);
    WeightArgs w;
    memset(&w, 0, sizeof(w));
    w.u0 = uniform;
    if (weights_in_rows) {
        w.logw_in = (const float*)d_all_rows + 7;
        w.in_stride = (int)(pack / 4);
        // 1: un-normalised weights (an update precedes: normalise first); 2: the shards' current normalised weights, as they are
        w.mode = (weights_in_rows == 2 ? 0 : WM_NORMALIZE) | WM_RESAMPLE_FORCE;
    } else {
        w.logw_in = f->logw_scratch;
        w.mode = WM_RESAMPLE_FORCE;
    }
    w.logw = f->logw_scratch;
    w.n = ng;
    w.n_new = ng;
    w.uniforms = f->d_uniforms;
    w.n_uniforms = 1;
    w.cdf = f->cdf;
    w.idx_out = f->idx;
    w.neff_out = f->neff;
    w.did_resample = f->did;
    w.n_weight_norm = ng;
    const int pnext = (f->pose_cur + (f->pose_for_update ? 2 : 1)) % 3;
    if (weights_in_rows == 1 && !idx_out && ng <= 1024) {
        // one launch: every import workgroup draws its own parent from the gathered weights (no weights launch to wait for)
        t_begin(f, PHD_K_WEIGHTS);
        HIPCHK(launch_gathered_resample(w, f->maps[f->cur ^ 1], f->counts[f->cur ^ 1], f->pose[pnext], d_all_rows, f->cap, pack,
                                        off, n, f->frozen ? nullptr : f->logw, (float)(-log((double)ng)),
                                        f->frozen ? nullptr : f->parent[f->pcur], f->cphd ? f->cn[f->cur ^ 1] : nullptr,
                                        f->cn_len, f->stream));
        t_end(f);
        f->rows_pending = false;
        if (!f->frozen) {
            f->cur ^= 1;
            f->pose_cur = (f->pose_cur + 1) % 3;
            f->parent_dirty = false;
        }
        return PHD_OK;
    }
    t_begin(f, PHD_K_WEIGHTS);
    HIPCHK(launch_weights_f(f, w));
    t_end(f);
    // copy_particles (src/slamtypes.h:313-333): slot j <- gathered row idx[off + j]; weights <- -log N and the map
    // indirection back to identity in the same launch
    HIPCHK(launch_import(f->maps[f->cur ^ 1], f->counts[f->cur ^ 1], f->pose[pnext], nullptr, d_all_rows, f->cap, pack, n,
                         f->stream, f->idx + off, f->frozen ? nullptr : f->logw, (float)(-log((double)ng)),
                         f->frozen ? nullptr : f->parent[f->pcur]));
    if (f->cphd)
        HIPCHK(launch_copy_rows((const float*)d_all_rows + 8 + 6 * f->cap, pack / 4, f->idx + off, nullptr, f->cn[f->cur ^ 1],
                                f->cn_len, nullptr, f->cn_len, n, f->stream));
    f->rows_pending = false;
    if (!f->frozen) { // bench protocol when frozen: the exchange ran, the snapshot stays
        f->cur ^= 1;
        f->pose_cur = (f->pose_cur + 1) % 3;
        f->parent_dirty = false;
    }
    if (idx_out) {
        HIPCHK(hipMemcpyAsync(idx_out, f->idx, (size_t)ng * sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    return PHD_OK;
}

// ---------------------------------------------------------------------------------------------
// inspection
// ---------------------------------------------------------------------------------------------
extern "C" int phd_debug_enable(phd_filter* f, int enable)
{
    CHECK_F(f);
    f->debug = (enable & 1) != 0;
    f->want_stamps = (enable & 2) != 0; // diagnostic instantiation with phase stamps
    f->fuse_enabled = (enable & 4) == 0; // bit 2: never fuse the weights routine into the update launch (parity tests)
    if (f->debug) return ensure_debug(f);
    return PHD_OK;
}

extern "C" int phd_debug_get_stamps(phd_filter* f, uint64_t* out)
{
    CHECK_F(f);
    if (!f->stamps || !out) return fail(PHD_ERR_INVALID_ARG, "phd_debug_get_stamps: enable stamps with phd_debug_enable(f, 2) and run an update first");
    HIPCHK(hipMemcpyAsync(out, f->stamps, ((size_t)f->n * PHD_STAMP_ROW + 8) * 8, hipMemcpyDeviceToHost, f->stream)); // + 8 stamps of the weights kernel
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

extern "C" int phd_debug_get_survivors(phd_filter* f, int particle, phd_gaussian2d* out, int32_t* slab_index_out,
                                       int capacity, int32_t* n_out)
{
    CHECK_F(f);
    if (!f->debug || !f->dbg_surv) return fail(PHD_ERR_INVALID_ARG, "phd_debug_get_survivors: call phd_debug_enable first");
    if (particle < 0 || particle >= f->n) return fail(PHD_ERR_INVALID_ARG, "bad particle index");
    int n = 0, nin = 0;
    HIPCHK(hipMemcpyAsync(&n, f->dbg_n + particle, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(&nin, f->dbg_nin + particle, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    const int dc = f->spill_cap ? f->spill_cap : f->S_cap;
    std::vector<float> s((size_t)6 * dc);
    std::vector<int> u(dc);
    HIPCHK(hipMemcpyAsync(s.data(), f->dbg_surv + (size_t)particle * 6 * dc, s.size() * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(u.data(), f->dbg_u + (size_t)particle * dc, u.size() * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (n_out) *n_out = n;
    if (!out) return PHD_OK;
    if (n > capacity) return fail(PHD_ERR_CAPACITY, "phd_debug_get_survivors: output buffer too small");
    // the kernel appends survivors in arrival order; the slab order is recovered from the slab index
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return u[a] < u[b]; });
    const int n_update = nin * (f->last_M + 1) + f->last_M;
    int n_near = 0;
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        out[k].weight = s[0 * dc + i];
        out[k].mean[0] = s[1 * dc + i];
        out[k].mean[1] = s[2 * dc + i];
        out[k].cov[0] = s[3 * dc + i];
        out[k].cov[1] = s[4 * dc + i];
        out[k].cov[2] = s[4 * dc + i];
        out[k].cov[3] = s[5 * dc + i];
        if (slab_index_out) slab_index_out[k] = (u[i] >= 0x40000000) ? (n_update + n_near++) : u[i];
    }
    return PHD_OK;
}

extern "C" int phd_debug_get_weight_increments(phd_filter* f, float* dlogw_out)
{
    CHECK_F(f);
    if (!dlogw_out) return fail(PHD_ERR_INVALID_ARG, "null output");
    HIPCHK(hipMemcpyAsync(dlogw_out, f->dlogw, f->n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return PHD_OK;
}

// the device's step report block -> host struct; interprets the sticky status bits.  A time-out of the fused step's
// weights workgroup left the ticket counter mid-count: the counter is re-zeroed and the bit cleared here, so the
// NEXT step is sound again — this one is reported as failed.
static int interpret_report(phd_filter* f, const unsigned raw[8], phd_step_report* out)
{
    phd_step_report r;
    r.status = raw[0];
    r.max_survivors = (int32_t)raw[1];
    r.max_map = (int32_t)raw[2];
    memcpy(&r.neff, &raw[3], 4);
    r.did_resample = (int32_t)raw[4];
    if (out) *out = r;
    if (r.status & PHD_STATUS_TAIL_TIMEOUT) {
        (void)hipStreamSynchronize(f->stream);
        // what the waiters saw (diagnostics): the hand-off ticket, the arrival counters of the block form's two barriers, its
        // leavers, and the code of the wait that gave up first (1: ticket, 2 / 3: first / second barrier; << 8: what it read)
        unsigned seen[5] = {0, 0, 0, 0, 0};
        (void)hipMemcpy(&seen[0], f->ticket, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&seen[1], f->gw_sync, 16, hipMemcpyDeviceToHost);
        char where[160];
        snprintf(where, sizeof(where), " [ticket %u, barrier arrivals %u / %u, leavers %u, first time-out: wait %u read %u]", seen[0], seen[1], seen[2],
                 seen[3], seen[4] & 255u, seen[4] >> 8);
        (void)hipMemsetAsync(f->ticket, 0, 4, f->stream);
        (void)hipMemsetAsync(f->gw_sync, 0, 16, f->stream);
        const unsigned cleared = r.status & ~(unsigned)PHD_STATUS_TAIL_TIMEOUT;
        (void)hipMemcpyAsync(f->status, &cleared, 4, hipMemcpyHostToDevice, f->stream);
        (void)hipStreamSynchronize(f->stream);
        return fail(PHD_ERR_HIP, std::string("fused step: the weights workgroup timed out waiting for the particles' workgroups "
                                             "(step failed; the hand-off counter has been reset)") + where);
    }
    if (r.status) return fail(PHD_ERR_CAPACITY, std::string("device capacity overflow:") + ((r.status & 1) ? " map_capacity" : "") +
                                                    ((r.status & 2) ? " survivor_capacity" : ""));
    return PHD_OK;
}

extern "C" int phd_step_report_get(phd_filter* f, phd_step_report* out)
{
    CHECK_F0(f);
    unsigned raw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(raw, f->report, sizeof(raw), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return interpret_report(f, raw, out);
}

extern "C" int phd_device_status(phd_filter* f, uint32_t* status_out, int32_t* max_survivors_out, int32_t* max_map_out)
{
    phd_step_report r;
    memset(&r, 0, sizeof(r));
    const int rc = phd_step_report_get(f, &r);
    if (status_out) *status_out = r.status;
    if (max_survivors_out) *max_survivors_out = r.max_survivors;
    if (max_map_out) *max_map_out = r.max_map;
    return rc;
}

// Declare that the filter now holds n particles (1 <= n <= its maximum: n_particles, or 5 n_particles
// n_predict_particles with the particle shotgun): the host-side mirror of a SynthSLAM whose particle count changed
// outside the library (src/phdfilter.cu:1185-1238 grows it in phdPredict, src/main.cpp:1289 shrinks it).  The map
// indirection is reset; the caller uploads particles and maps afterwards.
extern "C" int phd_set_particle_count(phd_filter* f, int n)
{
    CHECK_F(f);
    if (n < 1 || n > f->n_max) return fail(PHD_ERR_CAPACITY, "phd_set_particle_count: n outside 1..maximum particle count");
    if (f->frozen) return fail(PHD_ERR_UNSUPPORTED, "phd_set_particle_count: filter is frozen");
    f->n = n;
    HIPCHK(launch_iota(f->parent[f->pcur], f->n_max, f->stream));
    f->parent_dirty = false;
    f->pose_for_update = nullptr;
    return PHD_OK;
}
