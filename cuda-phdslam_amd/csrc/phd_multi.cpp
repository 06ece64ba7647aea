// phd_multi.cpp — one GM-PHD-SLAM filter sharded over the GPUs of one process: the C++ multi-device host of
// include/phdslam_multi.h over the single-shard C-ABI (include/phdslam.h) and RCCL.
//
// One host thread enqueues everything: per shard one HIP stream, per step
//   local step (ONE launch per shard) -> RCCL all-gather -> global normalise / indices (one launch per shard)
//   -> [index download, once] -> plan + export -> ncclSend/ncclRecv pairs -> import
// or, for small shards, local step into export rows -> ONE all-gather of whole shards -> normalise + indices + import.
// The reference has no counterpart (single GPU, src/main.cpp:1449); the loop body mirrored is src/main.cpp:1244-1297.
//
// Shards that share a device cannot form an RCCL communicator (one rank per GPU): they exchange by stream-ordered
// device copies behind the same two transport functions (all_gather, all_to_all) — what the one-GPU tests run.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "phdslam.h"
#include "phdslam_multi.h"

extern "C" int phd_internal_set_error(int code, const char* msg);

namespace {

int fail(int code, const std::string& msg) { return phd_internal_set_error(code, msg.c_str()); }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(PHD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCLCHK(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return fail(PHD_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define PHDCHK(expr)                                                                                   \
    do {                                                                                                \
        int rc_ = (expr);                                                                               \
        if (rc_ != PHD_OK) return rc_;                                                                  \
    } while (0)

struct Shard {
    int device = 0;
    phd_filter* f = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    float* allw = nullptr;              // [N] gathered raw log-weights
    unsigned char* recv = nullptr;      // [n][pack] all-to-all receive buffer
    unsigned char* allrows = nullptr;   // [N][pack] gathered shards (small-shard exchange), allocated on first use
    phd_measurement* d_z = nullptr;     // [MM] this step's scan
    phd_ackerman_noise* d_noise = nullptr; // [n] this step's control noise
    hipEvent_t ready = nullptr;         // peer copies: this shard's outgoing data is complete
    hipEvent_t done = nullptr;          //              this shard has consumed its peers' data
    bool done_pending = false;
    std::vector<int32_t> sc, rc;        // all-to-all counts
    void* send_buf = nullptr;
};

} // namespace

struct phd_multi {
    phd_slam_config cfg;
    int world = 0, N = 0, n = 0, cap = 0, MM = 0;
    int kpred = 1, n_max = 0;          // particle shotgun (n_predict_particles): every shard's set grows k-fold per predict, up to n_max
    size_t pack = 0, gathered_limit = 32u << 20;
    bool rccl = false, gathered = false, frozen = false;
    bool pull = false;                  // migrants are read straight out of the owners' slabs (phd_global_resample_pull)
    std::vector<Shard> sh;
    int32_t* h_idx = nullptr;           // pinned: the global resample indices (downloaded once per step from shard 0)
    int n_meas = 0;                     // of the resident inputs
    bool have_noise = false;
    // per-phase timing (phd_multi_timing_*): HIP events on shard 0's stream at the phase boundaries of a step
    bool scratch_current = false;       // every shard's logw_scratch holds the CURRENT normalised global weights (phd_global_normalize ran
                                        // and nothing has changed a weight or the particle count since)
    bool timing = false;
    // the longest step records start + local + gather + weights (normalise) + weights (indices) + plan + send/recv + import = 8
    // marks; sized with slack, and t_mark refuses (instead of dropping a span silently) beyond it
    static constexpr int MARKS = 12;
    hipEvent_t tev[MARKS] = {};
    int tmark = 0;                      // events recorded in the step in flight
    int tphase[MARKS] = {};
    double t_us[PHD_MULTI_PHASES] = {};
    int64_t t_steps = 0;
};

namespace {

// particles every shard holds NOW (n, or more between a shotgun predict and the resample that follows)
int ncur(const phd_multi* m) { return phd_n_particles(m->sh[0].f); }

// phase marks (timing pass only): an event on shard 0's stream; phase = what the span ENDING at this mark was
int t_mark(phd_multi* m, int phase)
{
    if (!m->timing) return PHD_OK;
    if (m->tmark >= phd_multi::MARKS) return fail(PHD_ERR_INVALID_ARG, "phd_multi: more phase marks in one step than the timing pass holds");
    Shard& s = m->sh[0];
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipEventRecord(m->tev[m->tmark], s.stream));
    m->tphase[m->tmark] = phase;
    m->tmark += 1;
    return PHD_OK;
}
// end of a step: all shards drained, spans accumulated (the timing pass synchronises every step; the timed loop does not run it)
int t_finish(phd_multi* m, bool count_step = true)
{
    if (!m->timing) return PHD_OK;
    for (auto& s : m->sh) {
        HIPCHK(hipSetDevice(s.device));
        HIPCHK(hipStreamSynchronize(s.stream));
    }
    for (int k = 1; k < m->tmark; ++k) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, m->tev[k - 1], m->tev[k]));
        m->t_us[m->tphase[k]] += 1e3 * (double)ms;
    }
    if (m->tmark > 1 && count_step) m->t_steps += 1;
    m->tmark = 0;
    return PHD_OK;
}

// stream-ordered hand-off for the peer-copy transport: every consumer stream waits for every producer's `ready`
int peer_publish(phd_multi* m)
{
    for (auto& s : m->sh) {
        HIPCHK(hipSetDevice(s.device));
        HIPCHK(hipEventRecord(s.ready, s.stream));
    }
    return PHD_OK;
}
int peer_consumed(phd_multi* m)
{
    for (auto& s : m->sh) {
        HIPCHK(hipSetDevice(s.device));
        HIPCHK(hipEventRecord(s.done, s.stream));
        s.done_pending = true;
    }
    return PHD_OK;
}
// before a shard overwrites data its peers read in the previous exchange
int peer_wait_consumed(phd_multi* m)
{
    if (m->rccl && !m->pull) return PHD_OK;            // (RCCL collectives order themselves; direct peer reads do not)
    for (auto& s : m->sh) {
        HIPCHK(hipSetDevice(s.device));
        for (auto& o : m->sh)
            if (&o != &s && o.done_pending) HIPCHK(hipStreamWaitEvent(s.stream, o.done, 0));
    }
    return PHD_OK;
}

// every shard's `count` bytes at src[k] -> dst[k] + j * count for all (k, j): dst of shard k = [world][count]
int all_gather(phd_multi* m, const std::vector<const void*>& src, const std::vector<void*>& dst, size_t bytes)
{
    if (m->rccl) {
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < m->world; ++k)
            NCCLCHK(ncclAllGather(src[k], dst[k], bytes, ncclChar, m->sh[k].comm, m->sh[k].stream));
        NCCLCHK(ncclGroupEnd());
        return PHD_OK;
    }
    PHDCHK(peer_publish(m));
    for (int k = 0; k < m->world; ++k) {
        Shard& s = m->sh[k];
        HIPCHK(hipSetDevice(s.device));
        for (int j = 0; j < m->world; ++j) {
            if (j != k) HIPCHK(hipStreamWaitEvent(s.stream, m->sh[j].ready, 0));
            if (j == k && src[j] == (unsigned char*)dst[k] + (size_t)j * bytes) continue; // in place
            HIPCHK(hipMemcpyAsync((unsigned char*)dst[k] + (size_t)j * bytes, src[j], bytes, hipMemcpyDeviceToDevice, s.stream));
        }
    }
    return peer_consumed(m);
}

// packed particles: shard j's send buffer is grouped by destination (ascending, self skipped), shard k's receive
// buffer by source (ascending, self skipped); recv_counts_k[j] == send_counts_j[k]
int all_to_all(phd_multi* m)
{
    const size_t pack = m->pack;
    if (m->rccl) {
        NCCLCHK(ncclGroupStart());
        for (int j = 0; j < m->world; ++j) {
            Shard& s = m->sh[j];
            size_t so = 0, ro = 0;
            for (int r = 0; r < m->world; ++r) {
                if (r == j) continue;
                if (s.sc[r]) NCCLCHK(ncclSend((const unsigned char*)s.send_buf + so * pack, (size_t)s.sc[r] * pack, ncclChar, r, s.comm, s.stream));
                if (s.rc[r]) NCCLCHK(ncclRecv(s.recv + ro * pack, (size_t)s.rc[r] * pack, ncclChar, r, s.comm, s.stream));
                so += s.sc[r];
                ro += s.rc[r];
            }
        }
        NCCLCHK(ncclGroupEnd());
        return PHD_OK;
    }
    PHDCHK(peer_publish(m));
    for (int k = 0; k < m->world; ++k) {
        Shard& d = m->sh[k];
        HIPCHK(hipSetDevice(d.device));
        size_t ro = 0;
        for (int j = 0; j < m->world; ++j) {
            if (j == k) continue;
            const Shard& s = m->sh[j];
            size_t so = 0;
            for (int r = 0; r < k; ++r)
                if (r != j) so += s.sc[r];
            if (d.rc[j]) {
                HIPCHK(hipStreamWaitEvent(d.stream, s.ready, 0));
                HIPCHK(hipMemcpyAsync(d.recv + ro * pack, (const unsigned char*)s.send_buf + so * pack, (size_t)d.rc[j] * pack,
                                      hipMemcpyDeviceToDevice, d.stream));
            }
            ro += d.rc[j];
        }
    }
    return peer_consumed(m);
}

int ensure_allrows(phd_multi* m)
{
    for (int k = 0; k < m->world; ++k) {
        Shard& s = m->sh[k];
        if (!s.allrows) {
            HIPCHK(hipSetDevice(s.device));
            HIPCHK(hipMalloc((void**)&s.allrows, (size_t)m->N * m->pack));
            // the fused step writes its rows straight into this shard's segment: the all-gather is in place (no self copy)
            PHDCHK(phd_set_rows_target(s.f, (unsigned char*)s.allrows + (size_t)k * m->n * m->pack));
        }
    }
    return PHD_OK;
}


// global resample + migration.  from_raw: the gathered UN-normalised weights sit in allw (forced resample straight after
// the update: normalisation and indices come from one launch); otherwise the indices come from the normalised global vector
// phd_global_normalize left on every shard.
int resample_stage(phd_multi* m, double uniform, bool from_raw)
{
    const int W = m->world;
    std::vector<const void*> src(W);
    std::vector<void*> dst(W);
    if (m->gathered && !from_raw) {
        // small shards: whole shards travel, nothing waits for the host
        PHDCHK(ensure_allrows(m));
        PHDCHK(peer_wait_consumed(m));
        // (no fresh global normalisation — resampleParticles without an update before it: the rows carry every shard's CURRENT
        //  normalised weights and the routine uses them as they are)
        const bool stale = !m->scratch_current;
        for (int k = 0; k < W; ++k) {
            void* rows = nullptr;
            if (stale) PHDCHK(phd_export_shard_current_dev(m->sh[k].f, &rows, nullptr));
            else PHDCHK(phd_export_shard_dev(m->sh[k].f, &rows, nullptr));
            src[k] = rows;
            dst[k] = m->sh[k].allrows;
        }
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_PLAN_EXPORT));     // (the export of the whole shard into its rows)
        PHDCHK(all_gather(m, src, dst, (size_t)m->n * m->pack));
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_ALL_GATHER));
        for (int k = 0; k < W; ++k)
            PHDCHK(phd_global_resample_gathered(m->sh[k].f, m->sh[k].allrows, uniform, W, k, stale ? 2 : 0, nullptr));
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_IMPORT));
        m->scratch_current = false;
        return PHD_OK;
    }
    // A resample that no update + global normalisation precedes (a control-only step of the particle shotgun crossing
    // 5 n_particles, src/main.cpp:1286; resampleParticles called twice): the shards' scratch copies of the global weights are
    // stale — shorter than the grown set, or from before the last resample.  Every shard's CURRENT normalised weights are
    // gathered instead (also the cross-device ordering point the PULL exchange needs) and used as they are.
    const bool regather = !from_raw && !m->scratch_current;
    if (regather) {
        PHDCHK(peer_wait_consumed(m));
        for (int k = 0; k < W; ++k) {
            float* lw = nullptr;
            PHDCHK(phd_logweights_dev(m->sh[k].f, &lw));
            src[k] = lw;
            dst[k] = m->sh[k].allw;
        }
        PHDCHK(all_gather(m, src, dst, (size_t)ncur(m) * sizeof(float)));
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_ALL_GATHER));
    }
    // indices on every shard (identical), downloaded ONCE; every shard plans from the same host copy
    int32_t* d_idx0 = nullptr;
    for (int k = 0; k < W; ++k) {
        int32_t* d_idx = nullptr;
        if (regather) PHDCHK(phd_global_resample_launch_normalized(m->sh[k].f, m->sh[k].allw, uniform, &d_idx));
        else PHDCHK(phd_global_resample_launch(m->sh[k].f, from_raw ? m->sh[k].allw : nullptr, uniform, &d_idx));
        if (k == 0) d_idx0 = d_idx;
    }
    m->scratch_current = false;                         // (the resample below changes every weight)
    PHDCHK(t_mark(m, PHD_MULTI_PHASE_WEIGHTS));
    if (m->pull) {
        // no host round trip: every shard copies its slots' parents straight out of the owners' slabs.  Every owner's update
        // is complete by now on every reader's stream (the all-gather of the log-weights that precedes this stage — an RCCL
        // collective, or the event waits of the device-copy transport); the readers' `done` events keep the owners' next
        // update from overwriting what is being read (peer_wait_consumed at the head of the next step).
        std::vector<phd_peer_view> views((size_t)W);
        for (int k = 0; k < W; ++k) PHDCHK(phd_peer_view_get(m->sh[k].f, &views[k]));
        for (int k = 0; k < W; ++k) PHDCHK(phd_global_resample_pull(m->sh[k].f, views.data(), W, k));
        PHDCHK(peer_consumed(m));
        return t_mark(m, PHD_MULTI_PHASE_IMPORT);
    }
    HIPCHK(hipSetDevice(m->sh[0].device));
    HIPCHK(hipMemcpyAsync(m->h_idx, d_idx0, (size_t)m->N * sizeof(int32_t), hipMemcpyDeviceToHost, m->sh[0].stream));
    HIPCHK(hipStreamSynchronize(m->sh[0].stream));
    PHDCHK(peer_wait_consumed(m));
    for (int k = 0; k < W; ++k)
        PHDCHK(phd_global_resample_plan(m->sh[k].f, m->h_idx, W, k, m->sh[k].sc.data(), m->sh[k].rc.data(), &m->sh[k].send_buf));
    PHDCHK(t_mark(m, PHD_MULTI_PHASE_PLAN_EXPORT));
    PHDCHK(all_to_all(m));
    PHDCHK(t_mark(m, PHD_MULTI_PHASE_SEND_RECV));
    for (int k = 0; k < W; ++k) PHDCHK(phd_global_resample_end(m->sh[k].f, m->sh[k].recv));
    PHDCHK(t_mark(m, PHD_MULTI_PHASE_IMPORT));
    return PHD_OK;
}

// local predict + update on every shard, the raw weights of all shards gathered on every shard
int update_stage(phd_multi* m, const phd_ackerman_control* u)
{
    const int W = m->world, M = m->n_meas;
    std::vector<const void*> src(W);
    std::vector<void*> dst(W);
    for (int k = 0; k < W; ++k) {
        if (u && m->kpred > 1) {
            // particle shotgun (src/phdfilter.cu:1185-1238): the staged calls — the predict multiplies every shard's set by k
            PHDCHK(phd_predict_ackerman_dev(m->sh[k].f, *u, m->have_noise ? m->sh[k].d_noise : nullptr));
            PHDCHK(phd_update_local_dev(m->sh[k].f, m->sh[k].d_z, M));
        }
        else if (u) PHDCHK(phd_step_local_dev(m->sh[k].f, *u, m->have_noise ? m->sh[k].d_noise : nullptr, m->sh[k].d_z, M));
        else PHDCHK(phd_update_local_dev(m->sh[k].f, m->sh[k].d_z, M));                // no motion (step 0, src/main.cpp:1244)
        float* raw = nullptr;
        PHDCHK(phd_raw_logweights_dev(m->sh[k].f, &raw));
        src[k] = raw;
        dst[k] = m->sh[k].allw;
    }
    PHDCHK(t_mark(m, PHD_MULTI_PHASE_LOCAL_STEP));
    PHDCHK(all_gather(m, src, dst, (size_t)ncur(m) * sizeof(float)));
    return t_mark(m, PHD_MULTI_PHASE_ALL_GATHER);
}

} // namespace

extern "C" int phd_multi_create(const phd_slam_config* cfg, const phd_multi_options* opt, phd_multi** out)
{
    if (!cfg || !out) return fail(PHD_ERR_INVALID_ARG, "phd_multi_create: null argument");
    phd_multi_options o;
    memset(&o, 0, sizeof(o));
    if (opt) o = *opt;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PHD_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
    const int world = o.n_shards > 0 ? o.n_shards : ndev;
    if (cfg->n_particles <= 0 || cfg->n_particles % world)
        return fail(PHD_ERR_INVALID_ARG, "phd_multi_create: n_particles must be a positive multiple of the shard count");
    phd_multi* m = new phd_multi();
    m->cfg = *cfg;
    m->world = world;
    m->N = cfg->n_particles;
    m->n = m->N / world;
    m->kpred = cfg->nPredictParticles > 1 ? cfg->nPredictParticles : 1;
    m->n_max = m->kpred > 1 ? 5 * m->n * m->kpred : m->n;        // as phd_create sizes a shard (src/main.cpp:1286: resample above 5 n)
    if (o.gathered_limit_bytes) m->gathered_limit = o.gathered_limit_bytes;
    m->sh.resize(world);
    bool distinct = true;
    for (int k = 0; k < world; ++k) {
        m->sh[k].device = o.devices ? o.devices[k] : k % ndev;
        if (m->sh[k].device < 0 || m->sh[k].device >= ndev) { delete m; return fail(PHD_ERR_INVALID_ARG, "phd_multi_create: bad device ordinal"); }
        for (int j = 0; j < k; ++j) distinct = distinct && m->sh[j].device != m->sh[k].device;
    }
    if (o.transport == PHD_TRANSPORT_RCCL && !distinct) { delete m; return fail(PHD_ERR_INVALID_ARG, "phd_multi_create: RCCL needs one device per shard"); }
    m->rccl = o.transport == PHD_TRANSPORT_RCCL || (o.transport == PHD_TRANSPORT_AUTO && distinct);
    int rc = PHD_OK;
    for (int k = 0; k < world && rc == PHD_OK; ++k) {
        Shard& s = m->sh[k];
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(PHD_ERR_HIP, "phd_multi_create: stream creation failed");
            break;
        }
        phd_options po;
        memset(&po, 0, sizeof(po));
        po.n_particles = m->n;
        po.map_capacity = o.map_capacity;
        po.max_measurements = o.max_measurements;
        po.survivor_capacity = o.survivor_capacity;
        po.device = s.device;
        po.stream = s.stream;
        po.global_particles = m->N;
        po.global_offset = k * m->n;
        rc = phd_create(cfg, &po, &s.f);
        if (rc) break;
        s.sc.assign(world, 0);
        s.rc.assign(world, 0);
    }
    if (rc == PHD_OK) {
        m->cap = phd_map_capacity(m->sh[0].f);
        m->MM = o.max_measurements > 0 ? std::min(o.max_measurements, PHD_MAX_MEASUREMENTS) : PHD_MAX_MEASUREMENTS;
        m->pack = phd_particle_pack_bytes(m->sh[0].f);
        // PHD_MULTI_EXCHANGE = gathered | pull | alltoall picks the form an AUTO create takes (an explicit option wins): a way to
        // fall back to the host-planned exchange on a machine whose peer reads misbehave, without rebuilding the caller
        int exchange = o.exchange;
        if (exchange == PHD_EXCHANGE_AUTO) {
            const char* ex = getenv("PHD_MULTI_EXCHANGE");
            if (ex && !strcmp(ex, "gathered")) exchange = PHD_EXCHANGE_GATHERED;
            else if (ex && !strcmp(ex, "pull")) exchange = PHD_EXCHANGE_PULL;
            else if (ex && !strcmp(ex, "alltoall")) exchange = PHD_EXCHANGE_ALLTOALL;
            else if (ex && *ex && strcmp(ex, "auto")) rc = fail(PHD_ERR_INVALID_ARG, std::string("PHD_MULTI_EXCHANGE=") + ex + ": expected gathered, pull, alltoall or auto");
        }
        m->gathered = exchange == PHD_EXCHANGE_GATHERED ||
                      (exchange == PHD_EXCHANGE_AUTO && (size_t)m->N * m->pack <= m->gathered_limit);
        // direct reads of the other shards' slabs: peer access between every pair of distinct devices
        // (PHD_MULTI_FLAG_NO_PEER_ACCESS: behave as if no pair of DISTINCT devices had it — the decision below under test;
        //  with every shard on one device it declares that one device "without peers" too, so that a one-GPU box can run it)
        bool peers = world <= 16 && !(o.flags & PHD_MULTI_FLAG_NO_PEER_ACCESS);
        for (int a = 0; a < world && peers; ++a)
            for (int b = 0; b < world && peers; ++b) {
                const int da = m->sh[a].device, db = m->sh[b].device;
                if (da == db) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { peers = false; break; }
                if (hipSetDevice(da) != hipSuccess) { peers = false; break; }
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peers = false;
                (void)hipGetLastError();
            }
        if (rc == PHD_OK && exchange == PHD_EXCHANGE_PULL && !peers)
            rc = fail(PHD_ERR_UNSUPPORTED, "phd_multi_create: PHD_EXCHANGE_PULL needs peer access between every pair of devices (at most 16 shards)");
        // AUTO without peer access: the host-planned exchange (index download + ncclSend/ncclRecv pairs) — never an error
        m->pull = peers && (exchange == PHD_EXCHANGE_PULL || exchange == PHD_EXCHANGE_AUTO);
        if (m->kpred > 1) {
            // the particle shotgun on shards: the grown set is resampled back by the PULL exchange (the other forms pack n particles)
            if (rc == PHD_OK && !m->pull) rc = fail(PHD_ERR_UNSUPPORTED, "phd_multi_create: n_predict_particles > 1 on shards needs the PULL exchange (peer access between the devices)");
            m->gathered = false;
        }
        for (auto& s : m->sh) {
            hipError_t e = hipSetDevice(s.device);
            if (e == hipSuccess) e = hipMalloc((void**)&s.allw, (size_t)world * m->n_max * sizeof(float));
            if (e == hipSuccess) e = hipMalloc((void**)&s.recv, (size_t)std::max(m->n, 1) * m->pack);
            if (e == hipSuccess) e = hipMalloc((void**)&s.d_z, (size_t)m->MM * sizeof(phd_measurement));
            if (e == hipSuccess) e = hipMalloc((void**)&s.d_noise, (size_t)m->n_max * sizeof(phd_ackerman_noise));
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ready, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
            // the local step writes its raw weights straight into this shard's segment of its own receive buffer: the all-gather of
            // the raw weights is in place (no self copy; on one rank nothing is left of it).  Not with the particle shotgun, whose
            // segments move with the particle count.
            if (e == hipSuccess && m->kpred == 1 && phd_set_raw_target(s.f, s.allw + (size_t)(&s - &m->sh[0]) * m->n) != PHD_OK) e = hipErrorUnknown;
            if (e != hipSuccess) { rc = fail(PHD_ERR_HIP, std::string("phd_multi_create: ") + hipGetErrorString(e)); break; }
        }
    }
    if (rc == PHD_OK && hipHostMalloc((void**)&m->h_idx, (size_t)m->N * sizeof(int32_t)) != hipSuccess)
        rc = fail(PHD_ERR_HIP, "phd_multi_create: pinned allocation failed");
    if (rc == PHD_OK && m->rccl) {
        std::vector<int> devs(world);
        std::vector<ncclComm_t> comms(world);
        for (int k = 0; k < world; ++k) devs[k] = m->sh[k].device;
        // one process, one node: the bootstrap needs no routable interface (a box without one may stall RCCL's interface
        // search).  The caller's own setting wins; ours is scoped to this call — the process environment is restored, so a
        // multi-node communicator the host creates later does not inherit a loopback-only bootstrap
        const bool had_if = getenv("NCCL_SOCKET_IFNAME") != nullptr;
        if (!had_if) setenv("NCCL_SOCKET_IFNAME", "lo", 1);
        // Communicator creation under a watchdog: a bootstrap that does not come back (seen once in round 2 on one box, before
        // the first output line of `phdslam --devices 1`) becomes an error with a pointer to the diagnostics instead of a
        // process that hangs forever.  PHD_RCCL_INIT_TIMEOUT = seconds (default 120; 0 = wait without limit).
        struct InitJob {
            std::vector<int> devs;
            std::vector<ncclComm_t> comms;
            ncclResult_t r = ncclSuccess;
            bool done = false;
            std::mutex mu;
            std::condition_variable cv;
        };
        auto job = std::make_shared<InitJob>();
        job->devs = devs;
        job->comms.resize(world);
        const char* te = getenv("PHD_RCCL_INIT_TIMEOUT");
        const int timeout_s = te ? atoi(te) : 120;
        std::thread([job, world]() {
            const ncclResult_t rr = ncclCommInitAll(job->comms.data(), world, job->devs.data());
            std::lock_guard<std::mutex> lk(job->mu);
            job->r = rr;
            job->done = true;
            job->cv.notify_all();
        }).detach();
        ncclResult_t r = ncclSuccess;
        bool timed_out = false;
        {
            std::unique_lock<std::mutex> lk(job->mu);
            if (timeout_s > 0) timed_out = !job->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return job->done; });
            else job->cv.wait(lk, [&] { return job->done; });
            if (!timed_out) { r = job->r; comms = job->comms; }
        }
        if (timed_out) {
            // The init thread is still inside ncclCommInitAll, on these devices, reading the environment: nothing it may touch is
            // torn down — the shards, their streams and the handle are LEAKED on purpose, NCCL_SOCKET_IFNAME stays as it is — and
            // the caller is told that this process cannot create another multi-device filter: it must exit (include/phdslam_multi.h)
            return fail(PHD_ERR_HIP, "ncclCommInitAll did not return within " + std::to_string(timeout_s) +
                                     " s (PHD_RCCL_INIT_TIMEOUT): run with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,BOOTSTRAP to see where "
                                     "the bootstrap waits; --shards on one device (device-copy transport) needs no communicator.  The "
                                     "bootstrap thread is still running: this process must exit (no further phd_multi_create)");
        }
        if (!had_if) unsetenv("NCCL_SOCKET_IFNAME");
        (void)hipSetDevice(m->sh[0].device);
        if (false) {}
        else if (r != ncclSuccess) rc = fail(PHD_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
        else for (int k = 0; k < world; ++k) m->sh[k].comm = comms[k];
    }
    if (rc != PHD_OK) {
        const std::string keep = phd_last_error();
        phd_multi_destroy(m);
        return fail(rc, keep);
    }
    *out = m;
    return PHD_OK;
}

extern "C" int phd_multi_destroy(phd_multi* m)
{
    if (!m) return PHD_OK;
    for (auto& s : m->sh) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.comm) (void)ncclCommDestroy(s.comm);
        if (s.f) phd_destroy(s.f);
        (void)hipFree(s.allw); (void)hipFree(s.recv); (void)hipFree(s.allrows); (void)hipFree(s.d_z); (void)hipFree(s.d_noise);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (m->h_idx) (void)hipHostFree(m->h_idx);
    for (auto& e : m->tev) if (e) (void)hipEventDestroy(e);
    delete m;
    return PHD_OK;
}

#define CHECK_M(m) do { if (!(m)) return fail(PHD_ERR_INVALID_ARG, "null multi-device handle"); } while (0)

extern "C" int phd_multi_n_shards(const phd_multi* m) { return m ? m->world : PHD_ERR_INVALID_ARG; }
extern "C" int phd_multi_n_particles(const phd_multi* m) { return m ? m->N : PHD_ERR_INVALID_ARG; }
extern "C" int phd_multi_n_particles_now(const phd_multi* m) { return m ? m->world * ncur(m) : PHD_ERR_INVALID_ARG; }
extern "C" int phd_multi_uses_rccl(const phd_multi* m) { return m ? (m->rccl ? 1 : 0) : PHD_ERR_INVALID_ARG; }
extern "C" int phd_multi_exchange_is_gathered(const phd_multi* m) { return m ? (m->gathered ? 1 : 0) : PHD_ERR_INVALID_ARG; }
extern "C" int phd_multi_exchange(const phd_multi* m)
{
    if (!m) return PHD_ERR_INVALID_ARG;
    return m->gathered ? PHD_EXCHANGE_GATHERED : (m->pull ? PHD_EXCHANGE_PULL : PHD_EXCHANGE_ALLTOALL);
}
extern "C" phd_filter* phd_multi_shard(phd_multi* m, int k) { return (m && k >= 0 && k < m->world) ? m->sh[k].f : nullptr; }

extern "C" int phd_multi_seed(phd_multi* m, uint64_t seed)
{
    CHECK_M(m);
    for (auto& s : m->sh) PHDCHK(phd_seed(s.f, seed));   // one stream of draws, indexed by the global particle index
    return PHD_OK;
}

extern "C" int phd_multi_set_config(phd_multi* m, const phd_slam_config* cfg)
{
    CHECK_M(m);
    if (!cfg) return fail(PHD_ERR_INVALID_ARG, "null config");
    if (cfg->n_particles != m->N) return fail(PHD_ERR_INVALID_ARG, "phd_multi_set_config: n_particles cannot change");
    for (auto& s : m->sh) PHDCHK(phd_set_config(s.f, cfg));
    m->cfg = *cfg;
    return PHD_OK;
}

extern "C" int phd_multi_set_frozen(phd_multi* m, int freeze)
{
    CHECK_M(m);
    for (auto& s : m->sh) PHDCHK(phd_set_frozen(s.f, freeze));
    m->frozen = freeze != 0;
    m->scratch_current = false;
    return PHD_OK;
}

extern "C" int phd_multi_sync(phd_multi* m)
{
    CHECK_M(m);
    for (auto& s : m->sh) PHDCHK(phd_sync(s.f));
    return PHD_OK;
}

extern "C" int phd_multi_set_particles(phd_multi* m, const phd_pose* poses, const float* log_weights, int n)
{
    CHECK_M(m);
    const int nc = ncur(m);
    if (n != m->world * nc) return fail(PHD_ERR_INVALID_ARG, "phd_multi_set_particles: n != the current particle count");
    m->scratch_current = false;
    for (int k = 0; k < m->world; ++k)
        PHDCHK(phd_set_particles(m->sh[k].f, poses ? poses + (size_t)k * nc : nullptr,
                                 log_weights ? log_weights + (size_t)k * nc : nullptr, nc));
    return PHD_OK;
}

extern "C" int phd_multi_get_particles(phd_multi* m, phd_pose* poses_out, float* log_weights_out)
{
    CHECK_M(m);
    const int nc = ncur(m);
    for (int k = 0; k < m->world; ++k)
        PHDCHK(phd_get_particles(m->sh[k].f, poses_out ? poses_out + (size_t)k * nc : nullptr,
                                 log_weights_out ? log_weights_out + (size_t)k * nc : nullptr));
    return PHD_OK;
}

extern "C" int phd_multi_set_maps(phd_multi* m, const phd_gaussian2d* concat, const int32_t* sizes)
{
    CHECK_M(m);
    if (!sizes) return fail(PHD_ERR_INVALID_ARG, "phd_multi_set_maps: null sizes");
    size_t off = 0;
    const int nc = ncur(m);
    for (int k = 0; k < m->world; ++k) {
        PHDCHK(phd_set_maps(m->sh[k].f, concat ? concat + off : nullptr, sizes + (size_t)k * nc));
        for (int p = 0; p < nc; ++p) off += (size_t)sizes[(size_t)k * nc + p];
    }
    return PHD_OK;
}

extern "C" int phd_multi_get_map_sizes(phd_multi* m, int32_t* sizes_out)
{
    CHECK_M(m);
    if (!sizes_out) return fail(PHD_ERR_INVALID_ARG, "null output");
    for (int k = 0; k < m->world; ++k) PHDCHK(phd_get_map_sizes(m->sh[k].f, sizes_out + (size_t)k * ncur(m)));
    return PHD_OK;
}

extern "C" int phd_multi_get_maps(phd_multi* m, phd_gaussian2d* concat_out, size_t concat_capacity, int32_t* sizes_out)
{
    CHECK_M(m);
    const int nc = ncur(m);
    std::vector<int32_t> sizes((size_t)m->world * nc);
    PHDCHK(phd_multi_get_map_sizes(m, sizes.data()));
    if (sizes_out) memcpy(sizes_out, sizes.data(), sizes.size() * sizeof(int32_t));
    if (!concat_out) return PHD_OK;
    size_t off = 0;
    for (int k = 0; k < m->world; ++k) {
        size_t tot = 0;
        for (int p = 0; p < nc; ++p) tot += (size_t)sizes[(size_t)k * nc + p];
        if (off + tot > concat_capacity) return fail(PHD_ERR_CAPACITY, "phd_multi_get_maps: output buffer too small");
        PHDCHK(phd_get_maps(m->sh[k].f, concat_out + off, concat_capacity - off, nullptr));
        off += tot;
    }
    return PHD_OK;
}

extern "C" int phd_multi_upload_inputs(phd_multi* m, const phd_ackerman_noise* noise, const phd_measurement* z, int n_meas)
{
    CHECK_M(m);
    const int M = std::min(std::max(n_meas, 0), m->MM);
    if (M > 0 && !z) return fail(PHD_ERR_INVALID_ARG, "phd_multi_upload_inputs: null measurements");
    PHDCHK(peer_wait_consumed(m));
    for (int k = 0; k < m->world; ++k) {
        Shard& s = m->sh[k];
        HIPCHK(hipSetDevice(s.device));
        if (M > 0) HIPCHK(hipMemcpyAsync(s.d_z, z, (size_t)M * sizeof(phd_measurement), hipMemcpyHostToDevice, s.stream));
        // (n_predict_particles = k: one draw per PREDICTED particle, k per prior particle, in global predicted order)
        const size_t nn = (size_t)ncur(m) * m->kpred;
        if (noise) HIPCHK(hipMemcpyAsync(s.d_noise, noise + (size_t)k * nn, nn * sizeof(phd_ackerman_noise), hipMemcpyHostToDevice, s.stream));
    }
    m->n_meas = M;
    m->have_noise = noise != nullptr;
    return PHD_OK;
}

// the step proper (phd_multi_step_resident wraps it with the phase-timing bookkeeping)
static int step_resident_body(phd_multi* m, phd_ackerman_control u, double uniform, int force_resample, int32_t* did_resample_out)
{
    const int W = m->world, M = m->n_meas;
    if (did_resample_out) *did_resample_out = 0;
    PHDCHK(peer_wait_consumed(m));
    PHDCHK(t_mark(m, 0));                               // start of the step
    m->scratch_current = false;
    if (M <= 0) {
        // no scan: predict only (src/main.cpp:1244-1260).  The nEff trigger needs measurements (:1286); a grown particle set
        // (n_predict_particles > 1) beyond 5 n_particles is resampled back all the same, and so is a forced step — what
        // phd_step_dev does on a single filter
        for (auto& s : m->sh) {
            if (m->kpred > 1) PHDCHK(phd_predict_ackerman_dev(s.f, u, m->have_noise ? s.d_noise : nullptr));
            else PHDCHK(phd_step_local_dev(s.f, u, m->have_noise ? s.d_noise : nullptr, s.d_z, 0));
        }
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_LOCAL_STEP));
        if (force_resample || W * ncur(m) > 5 * m->N) {
            if (did_resample_out) *did_resample_out = 1;
            return resample_stage(m, uniform, false);
        }
        return PHD_OK;
    }
    if (m->gathered && force_resample) {
        // small shards, forced resample: nothing waits for the host
        PHDCHK(ensure_allrows(m));
        std::vector<const void*> src(W);
        std::vector<void*> dst(W);
        for (int k = 0; k < W; ++k) {
            void* rows = nullptr;
            PHDCHK(phd_step_local_rows_dev(m->sh[k].f, u, m->have_noise ? m->sh[k].d_noise : nullptr, m->sh[k].d_z, M, &rows, nullptr));
            src[k] = rows;
            dst[k] = m->sh[k].allrows;
        }
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_LOCAL_STEP));
        PHDCHK(all_gather(m, src, dst, (size_t)m->n * m->pack));
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_ALL_GATHER));
        for (int k = 0; k < W; ++k)
            PHDCHK(phd_global_resample_gathered(m->sh[k].f, m->sh[k].allrows, uniform, W, k, 1, nullptr));
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_IMPORT));
        if (did_resample_out) *did_resample_out = 1;
        return PHD_OK;
    }
    // local step, then the raw weights of all shards on every shard
    PHDCHK(update_stage(m, &u));
    bool resample = force_resample != 0;
    if (!resample && m->pull && !m->gathered && m->kpred == 1 && !getenv("PHD_MULTI_HOST_NEFF")) {
        // The nEff trigger with no host round trip: every shard normalises, takes nEff and the decision on the device and
        // leaves the indices or the identity; the pull is enqueued either way (nothing moves when nothing was resampled: the
        // copy-free form).  The decision is downloaded only for a caller that asks for it — after everything is enqueued.
        bool ok = true;
        for (int k = 0; k < W; ++k) ok = ok && phd_global_resample_auto_supported(m->sh[k].f, W) == 1;
        if (ok) {
            for (int k = 0; k < W; ++k) PHDCHK(phd_global_resample_launch_auto(m->sh[k].f, m->sh[k].allw, uniform));
            PHDCHK(t_mark(m, PHD_MULTI_PHASE_WEIGHTS));
            std::vector<phd_peer_view> views((size_t)W);
            for (int k = 0; k < W; ++k) PHDCHK(phd_peer_view_get(m->sh[k].f, &views[k]));
            for (int k = 0; k < W; ++k) PHDCHK(phd_global_resample_pull_auto(m->sh[k].f, views.data(), W, k));
            PHDCHK(peer_consumed(m));
            PHDCHK(t_mark(m, PHD_MULTI_PHASE_IMPORT));
            m->scratch_current = false;                 // (the host does not know whether the weights were reset)
            if (did_resample_out) {
                phd_step_report r;
                memset(&r, 0, sizeof(r));
                (void)phd_step_report_get(m->sh[0].f, &r);      // (a status bit is the caller's to find, as on the other paths)
                *did_resample_out = r.did_resample;
            }
            return PHD_OK;
        }
    }
    if (!resample) {
        // the reference's trigger: global normalisation (adopted by every shard), nEff read back from shard 0 only
        float neff = 0.f;
        const int n_now = W * ncur(m);
        for (int k = W - 1; k >= 0; --k)
            PHDCHK(phd_global_normalize(m->sh[k].f, m->sh[k].allw, n_now, k == 0 ? &neff : nullptr));
        m->scratch_current = true;
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_WEIGHTS));
        resample = neff <= m->cfg.resampleThresh || n_now > 5 * m->N;                // src/main.cpp:1286
        if (!resample) return PHD_OK;
    }
    if (did_resample_out) *did_resample_out = 1;
    return resample_stage(m, uniform, force_resample != 0);
}

extern "C" int phd_multi_step_resident(phd_multi* m, phd_ackerman_control u, double uniform, int force_resample,
                                       int32_t* did_resample_out)
{
    CHECK_M(m);
    m->tmark = 0;
    const int rc = step_resident_body(m, u, uniform, force_resample, did_resample_out);
    if (rc != PHD_OK) return rc;
    return t_finish(m);
}

extern "C" int phd_multi_timing_enable(phd_multi* m, int enable)
{
    CHECK_M(m);
    if (enable && !m->tev[0]) {
        HIPCHK(hipSetDevice(m->sh[0].device));
        for (int k = 0; k < phd_multi::MARKS; ++k) HIPCHK(hipEventCreate(&m->tev[k]));
    }
    m->timing = enable != 0;
    m->tmark = 0;
    return PHD_OK;
}

extern "C" int phd_multi_timing_reset(phd_multi* m)
{
    CHECK_M(m);
    for (int k = 0; k < PHD_MULTI_PHASES; ++k) m->t_us[k] = 0.0;
    m->t_steps = 0;
    return PHD_OK;
}

extern "C" int phd_multi_timing_read(phd_multi* m, double* us_total, int64_t* steps_out)
{
    CHECK_M(m);
    if (!us_total || !steps_out) return fail(PHD_ERR_INVALID_ARG, "phd_multi_timing_read: null output");
    for (int k = 0; k < PHD_MULTI_PHASES; ++k) us_total[k] = m->t_us[k];
    *steps_out = m->t_steps;
    return PHD_OK;
}

extern "C" int phd_multi_step(phd_multi* m, phd_ackerman_control u, const phd_ackerman_noise* noise, const phd_measurement* z,
                              int n_meas, double uniform, int force_resample, int32_t* did_resample_out)
{
    CHECK_M(m);
    PHDCHK(phd_multi_upload_inputs(m, noise, z, n_meas));
    return phd_multi_step_resident(m, u, uniform, force_resample, did_resample_out);
}

extern "C" int phd_multi_state_snapshot(phd_multi* m, phd_pose* expected_out, phd_gaussian2d* map_out, int capacity,
                                        int32_t* n_map_out, int32_t* particle_out, phd_pose* poses_out, float* log_weights_out,
                                        phd_step_report* report_out)
{
    CHECK_M(m);
    if (!expected_out || !map_out || !n_map_out) return fail(PHD_ERR_INVALID_ARG, "phd_multi_state_snapshot: null output");
    std::vector<phd_pose> pp;
    std::vector<float> lw;
    phd_pose* poses = poses_out;
    float* logw = log_weights_out;
    const int nc = ncur(m), Nc = m->world * nc;
    if (!poses) { pp.resize((size_t)Nc); poses = pp.data(); }
    if (!logw) { lw.resize((size_t)Nc); logw = lw.data(); }
    PHDCHK(phd_multi_get_particles(m, poses, logw));
    // src/main.cpp:331-356: weighted-mean pose, arg-max weight (first maximum)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    float best = -3.4028235e38f;
    int bi = -1;
    for (int i = 0; i < Nc; ++i) {
        const double w = exp((double)logw[i]);
        acc[0] += w * poses[i].px; acc[1] += w * poses[i].py; acc[2] += w * poses[i].ptheta;
        acc[3] += w * poses[i].vx; acc[4] += w * poses[i].vy; acc[5] += w * poses[i].vtheta;
        if (logw[i] > best) { best = logw[i]; bi = i; }
    }
    if (bi < 0) return fail(PHD_ERR_NAN, "no finite particle weight");
    if (Nc == 1) *expected_out = poses[0];                                           // src/main.cpp:381-384
    else {
        expected_out->px = (float)acc[0]; expected_out->py = (float)acc[1]; expected_out->ptheta = (float)acc[2];
        expected_out->vx = (float)acc[3]; expected_out->vy = (float)acc[4]; expected_out->vtheta = (float)acc[5];
    }
    if (particle_out) *particle_out = bi;
    PHDCHK(phd_get_map(m->sh[bi / nc].f, bi % nc, map_out, capacity, n_map_out));
    if (report_out) {
        phd_step_report tot;
        memset(&tot, 0, sizeof(tot));
        int rc_all = PHD_OK;
        for (int k = 0; k < m->world; ++k) {
            phd_step_report r;
            memset(&r, 0, sizeof(r));
            const int rc = phd_step_report_get(m->sh[k].f, &r);
            if (rc != PHD_OK && rc_all == PHD_OK) rc_all = rc;
            tot.status |= r.status;
            tot.max_survivors = std::max(tot.max_survivors, r.max_survivors);
            tot.max_map = std::max(tot.max_map, r.max_map);
            if (k == 0) { tot.neff = r.neff; tot.did_resample = r.did_resample; }
        }
        *report_out = tot;
        if (rc_all != PHD_OK) return rc_all;
        if (tot.neff != tot.neff) return fail(PHD_ERR_NAN, "nan weights detected");
    }
    return PHD_OK;
}

extern "C" int phd_multi_expected_map(phd_multi* m, phd_gaussian2d* out, int capacity, int32_t* n_out)
{
    CHECK_M(m);
    // every shard concatenates its weighted maps on its device; shard 0 collects the planes in global particle order
    std::vector<float*> planes((size_t)m->world, nullptr);
    std::vector<int64_t> tot((size_t)m->world, 0);
    int64_t T = 0;
    for (int k = 0; k < m->world; ++k) {
        PHDCHK(phd_expected_map_concat_dev(m->sh[k].f, &planes[k], &tot[k]));     // synchronises shard k's stream
        T += tot[k];
    }
    if (T == 0) { if (n_out) *n_out = 0; return PHD_OK; }
    if (m->world == 1) return phd_gm_reduce_dev(m->sh[0].f, planes[0], T, 6, m->cfg.minSeparation, out, capacity, n_out);
    Shard& s0 = m->sh[0];
    HIPCHK(hipSetDevice(s0.device));
    float* all = nullptr;
    HIPCHK(hipMalloc((void**)&all, (size_t)6 * T * sizeof(float)));
    int64_t off = 0;
    hipError_t e = hipSuccess;
    for (int k = 0; k < m->world && e == hipSuccess; ++k) {
        for (int pl = 0; pl < 6 && e == hipSuccess && tot[k] > 0; ++pl)
            e = hipMemcpyPeerAsync(all + (size_t)pl * T + off, s0.device, planes[k] + (size_t)pl * tot[k], m->sh[k].device,
                                   (size_t)tot[k] * sizeof(float), s0.stream);
        off += tot[k];
    }
    int rc = e == hipSuccess ? phd_gm_reduce_dev(s0.f, all, T, 6, m->cfg.minSeparation, out, capacity, n_out)
                             : fail(PHD_ERR_HIP, hipGetErrorString(e));
    (void)hipStreamSynchronize(s0.stream);
    (void)hipFree(all);
    return rc;
}

// run_synth's order with the state log between update and resample (src/main.cpp:1260-1297): phd_multi_update = phdPredict +
// phdUpdateSynth + the global weight normalisation (every shard adopts its slice; the global nEff lands in the step report);
// phd_multi_resample = resampleParticles over the global set (the caller decides: nEff <= threshold and the step had
// measurements, :1286)
// The staged calls keep their own phase marks: each call starts from an empty mark list and closes it (the timing pass
// synchronises, like phd_multi_step_resident's), so a run of update / resample calls with timing on never fills the list.
// A step is counted by the update; the resample of the same step only adds its spans.
extern "C" int phd_multi_update(phd_multi* m, const phd_ackerman_control* u, const phd_ackerman_noise* noise,
                                const phd_measurement* z, int n_meas)
{
    CHECK_M(m);
    PHDCHK(phd_multi_upload_inputs(m, noise, z, n_meas));
    PHDCHK(peer_wait_consumed(m));
    m->tmark = 0;
    PHDCHK(t_mark(m, 0));
    m->scratch_current = false;
    if (m->n_meas <= 0) {
        if (u) for (auto& s : m->sh) {
            if (m->kpred > 1) PHDCHK(phd_predict_ackerman_dev(s.f, *u, m->have_noise ? s.d_noise : nullptr));
            else PHDCHK(phd_step_local_dev(s.f, *u, m->have_noise ? s.d_noise : nullptr, s.d_z, 0));
        }
        PHDCHK(t_mark(m, PHD_MULTI_PHASE_LOCAL_STEP));
        return t_finish(m);
    }
    PHDCHK(update_stage(m, u));
    for (int k = m->world - 1; k >= 0; --k) PHDCHK(phd_global_normalize(m->sh[k].f, m->sh[k].allw, m->world * ncur(m), nullptr));
    m->scratch_current = true;
    PHDCHK(t_mark(m, PHD_MULTI_PHASE_WEIGHTS));
    return t_finish(m);
}

extern "C" int phd_multi_resample(phd_multi* m, double uniform)
{
    CHECK_M(m);
    m->tmark = 0;
    PHDCHK(t_mark(m, 0));
    const int rc = resample_stage(m, uniform, false);
    if (rc != PHD_OK) { m->tmark = 0; return rc; }
    return t_finish(m, false);
}
