// phd_device.h — internal interface between the C-ABI host code (phd_api.cpp) and the gfx950
// kernels (phd_kernels.hip).  Not installed; the public boundary is include/phdslam.h.
#ifndef PHD_DEVICE_H
#define PHD_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phdslam.h"

namespace phd {

// the SlamConfig fields that reach the device (reference: __constant__ dev_config, the whole
// 324-byte struct, src/phdfilter.cu:121,3885-3890) — passed as a kernel argument (SGPRs)
struct DevConfig {
    float dt;
    float minRange, maxRange, maxBearing;
    float stdRange, stdBearing;
    float clutterDensity, pd;
    float clutterRate;          // CPHD only: Poisson clutter cardinality
    float birthWeight, birthNoiseFactor;
    float minFeatureWeight, minSeparation;
    float l, h, a, b;
    float stdAlpha, stdEncoder;
    int subdividePredict;
    int distanceMetric;
    int labeledMeasurements;
    int particleOffset;         // index of this shard's first particle in the global ordering (device noise generator)
};

#define PHD_STAMP_ROW 32   // u64 phase stamps per particle of the diagnostic instantiation
enum { PHD_STATUS_MAP_OVERFLOW = 1u, PHD_STATUS_SURVIVOR_OVERFLOW = 2u, PHD_STATUS_TAIL_TIMEOUT = 4u };

// Map slab layout in HBM: particle-major, then 6 SoA planes of `cap` floats:
//   slab(p) = base + p*6*cap ; planes: 0 weight, 1 mean x, 2 mean y, 3 cov xx, 4 cov xy, 5 cov yy
struct WeightArgs {
    const float* logw_in;       // [n] (element i at logw_in[i * in_stride] when in_stride != 0)
    int in_stride;              // floats between consecutive weights of logw_in (0: contiguous)
    float* logw;                // [n] working/out vector (== logw_in unless frozen)
    const float* dlogw;         // [n]
    float* raw_out;             // [n] optional: un-normalised accumulated weights
    int n;
    int n_new;
    int mode;
    float resample_thresh;
    const double* uniforms;     // device, n_uniforms entries (stratified); unused when n_uniforms == 1
    double u0;                  // the single uniform of systematic resampling (n_uniforms == 1)
    int n_uniforms;
    double* cdf;                // [n] scratch (used when n > 2048)
    int* idx_out;               // [n_new]
    float* neff_out;            // [1]
    int* did_resample;          // [1]
    // commit (copy_particles)
    const phd_pose* pose_in;
    phd_pose* pose_out;
    const int* parent_in;
    int* parent_out;
    int n_weight_norm;
    unsigned long long* wstamps; // [8] phase stamps (diagnostics) or NULL
    // block form of the routine (n > 4096, phd_weights.h: weights_grid_body): arrival counters of its grid-wide barriers
    // ([0], [1]: the two barriers, [2]: leavers; zero between launches) and one record of 8 words per block of 256 weights
    unsigned* gsync;
    float* gpart;
};

struct UpdateArgs {
    const float* map_in;
    const int* count_in;
    float* map_out;
    int* count_out;
    // rows mode (multi-GPU whole-shard exchange), out_stride != 0: the outputs of particle p land directly in its export
    // row of out_stride floats — slab at map_out + p * out_stride, header [pose | count | raw log-weight] in the 8 floats
    // before it, CPHD cardinality row at cn_out + p * out_stride.  0: slabs of 6 * cap floats, cn rows of cn_len.
    unsigned out_stride;
    const int* parent;          // map indirection left by the last resample
    int* parent_reset;          // == parent unless frozen (NULL): parent[p] <- p once slab p is rewritten
    const phd_pose* pose;       // [n_particles]
    const phd_measurement* z;   // [M] device
    float* dlogw;               // [n_particles] out
    const float* logw_in;       // [n_particles] current log-weights (read only when raw_out is set)
    float* raw_out;             // optional: raw_out[p] = logw_in[p] + dlogw[p] (multi-GPU step: feeds the all-gather)
    int M;
    int MM;                     // max measurements (LDS sizing)
    int cap;
    int S_cap;
    // inspection (NULL when disabled)
    float* dbg_surv;            // [n][6][dbg_cap]
    int* dbg_u;                 // [n][dbg_cap]
    int* dbg_n;                 // [n]
    int* dbg_nin;               // [n]
    int dbg_cap;                // survivors per particle the inspection buffers hold (S_cap, or the spill capacity)
    unsigned long long* stamps; // [n][PHD_STAMP_ROW] phase stamps (diagnostic instantiation) or NULL
    // fused vehicle predict (phd_step_dev): pose <- f(pose, control, noise) before the update
    int do_predict;
    phd_ackerman_control control;
    const phd_ackerman_noise* noise; // NULL: draw from (seed, counter)
    phd_pose* pose_out;
    unsigned long long seed, counter;
    // fused weights / resample tail (small particle counts): run by the last workgroup to finish
    int fuse_weights;
    unsigned* ticket;
#ifdef PHD_EXP_TRACE
    unsigned long long* trace;  // experiment: [grid][8] (start, end, HW_ID, hand-off, pass 2 + barrier, merge, pass 2) in 100 MHz ticks
#endif
    WeightArgs wa;
    unsigned* status;
    int* max_surv;
    int* max_map;
    // CPHD variant (filter_type = 1): per-particle log cardinality, rows of cn_len floats indexed like the slabs
    int cphd;
    const float* cn_in;
    float* cn_out;
    int cn_len;
    const float* lfact;         // log factorials 0..lfact_len-1 (initCphdConstants, src/phdfilter.cu.bak:421-425)
    int lfact_len;
    float2* cphd_scratch;       // [n][MM][MM] (mantissa, exponent) rows of the ESF backward sweep
    // spill path (survivor lists longer than S_cap; NULL / 0 when not enabled): see phd_spill.h
    float* spill_rec;           // [n][2][spill_cap][8] survivor records, and the sorted copy phd_merge_spill_kernel works on
    int* spill_meta;            // [n][8]: survivor count (0: the LDS merge handled the particle), n_update, n_out0, r1, input slab
    unsigned short* spill_out;  // [n][cap] indices of the untouched out-of-range features
    int spill_cap;              // records per particle
    int* spill_tmp;             // [n][spill_cap]: bucket member lists of the spill merge's sort
    long long* spill_acc;       // [n][cap][8]: per cluster six 64-bit words of exact moment sums + the seed's record (16 B)
    DevConfig cfg;
};

size_t update_lds_bytes(int S, int C, int MM);
size_t update_static_lds_bytes();           // static __shared__ of the update kernel's instantiations (needs a device)
size_t cphd_lds_bytes(int S_cap, int cn_len, int MM);   // extra LDS of the CPHD instantiation, carved after update_lds_bytes()
// rows of `len` floats: dst[(c ? c[k] : k)] = src[b ? b[a ? a[k] : k] : (a ? a[k] : k)], k < n; skipped when a[k] < 0
hipError_t launch_copy_rows(const float* src, size_t src_stride, const int* a, const int* b, float* dst, size_t dst_stride,
                            const int* c, int len, int n, hipStream_t st);
int update_fuse_max_particles();
int weights_grid_min_particles();          // particle counts above this take the block form of the weights routine
int weights_grid_workgroups(int n);        // workgroups it runs on
size_t weights_grid_lds_bytes(int n);      // LDS it needs (block ends of the CDF)

// three_per_cu: the filter's build, decided once at phd_create by update_takes_three_per_cu() (the 80-register instantiations)
bool update_takes_three_per_cu(bool cphd, bool spill, size_t lds_bytes, int n_base);
// any_layout: never the instantiations with a compiled-in LDS layout (phd_kernels.hip, LAYOUT), whatever the filter's layout is
hipError_t launch_update_merge(const UpdateArgs& a, int n_particles, size_t lds_bytes, hipStream_t st, bool three_per_cu, bool any_layout = false);
int update_workgroups_per_cu(const UpdateArgs& a, size_t lds_bytes, bool three_per_cu, bool any_layout = false);
int update_instantiation(const UpdateArgs& a, int n_particles, bool three_per_cu, bool any_layout);   // index into the launcher's table (diagnostics)   // what the runtime says a CU holds (-1: query failed)
#define PHD_MAX_PEERS 16        // shards whose memory one pull kernel can read (phd_global_resample_pull)
hipError_t launch_resample_pull(const phd_peer_view* views, int world, const int* idx, int off, int n_src, int n_dst, int rank, float* dst,
                                int* counts_dst, phd_pose* pose_dst, int cap, float* logw_fill, float nlw, int* parent_next,
                                float* cn_dst, int cn_len, hipStream_t st);
// the copy-free forms (phd_kernels.hip): local parents by indirection, remote ones into the guest slabs; goff = the first guest
// slab counted from the filter's current buffer
hipError_t launch_resample_pull_free(const phd_peer_view* views, int world, const int* idx, int off, int n_src, int n_dst, int rank,
                                     float* guests, int* counts_g, float* cn_g, int goff, phd_pose* pose_dst, int cap, float* logw_fill,
                                     float nlw, int* parent_next, int cn_len, hipStream_t st, const int* did = nullptr,
                                     const float* logw_keep = nullptr);
hipError_t launch_resample_end_free(const int* parent, const phd_pose* pose_src, const int* plan, int n, const void* recv, size_t stride,
                                    float* guests, int* counts_g, float* cn_g, int goff, phd_pose* pose_dst, int cap, float* logw_fill,
                                    float nlw, int* parent_next, int cn_len, hipStream_t st);
hipError_t launch_merge_spill(const UpdateArgs& a, int n_particles, hipStream_t st);   // no-op unless a.spill_rec
hipError_t launch_predict(const phd_pose* in, phd_pose* out, int n, phd_ackerman_control u,
                          const phd_ackerman_noise* noise, uint64_t seed, uint64_t counter, const DevConfig& cfg,
                          hipStream_t st);
hipError_t launch_predict_shotgun(const phd_pose* in, phd_pose* out, int n_pred, int k, phd_ackerman_control u,
                                  const phd_ackerman_noise* noise, uint64_t seed, uint64_t counter, const DevConfig& cfg,
                                  const int* parent_in, int* parent_out, const float* logw_in, float* logw_out,
                                  hipStream_t st);
hipError_t launch_weights(const WeightArgs& a, hipStream_t st, unsigned* status);   // status: the filter's status word (time-out bit of the block form)
hipError_t launch_pack_maps(const phd_gaussian2d* concat, const int* offsets, const int* sizes, float* slabs, int cap,
                            int n, hipStream_t st);
hipError_t launch_unpack_maps(const float* slabs, const int* parent, const int* offsets, const int* counts,
                              phd_gaussian2d* concat, int cap, int n, hipStream_t st);
hipError_t launch_unpack_one(const float* slabs, const int* parent, const int* counts, const int* which, phd_gaussian2d* out,
                             int cap, int* n_out, hipStream_t st);
hipError_t launch_state(const phd_pose* poses, const float* logw, int n, float* pose_out, int* argmax_out,
                        hipStream_t st);
hipError_t launch_export(const float* slabs, const int* counts, const int* parent, const phd_pose* poses,
                         const int* which, void* buf, int cap, size_t stride, int n, hipStream_t st,
                         const float* raw = nullptr);
hipError_t launch_gathered_resample(const WeightArgs& a, float* slabs, int* counts, phd_pose* poses, const void* rows, int cap,
                                    size_t stride, int off, int n, float* logw_fill, float nlw, int* parent_reset,
                                    float* cn_dst, int cn_len, hipStream_t st);
hipError_t launch_import(float* slabs, int* counts, phd_pose* poses, const int* which, const void* buf, int cap,
                         size_t stride, int n, hipStream_t st, const int* rowsel = nullptr, float* logw_fill = nullptr,
                         float nlw = 0.f, int* parent_reset = nullptr);
hipError_t launch_resample_end(const float* src, const int* counts_src, const int* parent, const phd_pose* pose_src, const int* plan,
                               int n, const void* recv, size_t stride, float* dst, int* counts_dst, phd_pose* pose_dst, int cap,
                               float* logw_fill, float nlw, int* parent_next, const float* cn_src, float* cn_dst, int cn_len,
                               hipStream_t st);
hipError_t launch_gather_maps(const float* src, const int* counts_src, const int* parent, const int* sel, float* dst,
                              int* counts_dst, const phd_pose* pose_src, phd_pose* pose_dst, int cap, int n,
                              hipStream_t st);
hipError_t launch_iota(int* a, int n, hipStream_t st);
// the pipelined state snapshot (phd_snapshot.hip): words of the staging block before the map
#define PHD_SNAP_HEADER_WORDS 24
hipError_t launch_snapshot(const phd_pose* poses, const float* logw, int n, int n_max, const float* slabs, const int* parent,
                           const int* counts, int cap, const float* cn, int cn_len, float* out, hipStream_t st);
hipError_t launch_fill(float* a, float v, int n, hipStream_t st);

// expected-a-posteriori map / device-wide reduceGaussianMixture (phd_eap.hip)
struct GmWorkspace;
GmWorkspace* gm_workspace_create();
void gm_workspace_destroy(GmWorkspace* w);
float* gm_concat_buffer(GmWorkspace* w, size_t T, hipStream_t st);  // [6][T] planes, grown on demand
int* gm_offsets_buffer(GmWorkspace* w, size_t n, hipStream_t st);
hipError_t launch_eap_concat(const float* maps, const int* counts, const int* parent, const float* logw, const int* offsets,
                             int cap, int n, float* out, size_t T, hipStream_t st);
// planes in[0..6] = weight, mean x, mean y, cov(0,0), cov(1,0), cov(0,1), cov(1,1); in[5] == in[4] when symmetric
hipError_t gm_reduce_device(GmWorkspace* w, const float* const in[7], size_t T, float min_distance, hipStream_t st,
                            int* K_out, int* rounds_out, const phd_gaussian2d** d_result);

// weight-kernel mode bits (mirrors the enum in phd_kernels.hip)
enum { WM_ACCUMULATE = 1, WM_NORMALIZE = 2, WM_RESAMPLE_FORCE = 4, WM_RESAMPLE_AUTO = 8, WM_HAD_MEAS = 16, WM_COMMIT = 32 };

} // namespace phd
#endif
