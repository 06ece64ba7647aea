/*
 * scphd_cpu.h — CPU ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the reference's Rao-Blackwellised GM-PHD-SLAM hot path
 * (cheesinglee/cuda-PHDSLAM: src/phdfilter.cu kernels + src/gm_reduce.cpp + the resampling of
 * src/main.cpp).  The reference's own src/scphd_cpu.cpp is a one-line stub, so
 * this file is the CPU statement of the algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in
 * oracle/.  The product (cuda-phdslam_amd/) never includes, links or calls it.
 *
 * PARITY PIN STATUS: pinned against the reference's executable artefacts for
 *   - Ackerman predict  (matlab/simData2_ackerman.mat sim.traj/sim.control under
 *                        python/AckermanMotionModel.py)            -> tests/golden/ackerman_kat.npz
 *   - predicted (r,b), in-range test, inverse measurement
 *                       (python/RangeBearingMeasurementModel.py)   -> tests/golden/rb_model_kat.npz
 *   - measurement loader (sim.data(k).measurements vs the text file)
 * Everything else (EKF gain/covariance, PHD weights, particle weight, prune, merge, resample)
 * has no reference-executable pin — the reference ships no tests, no outputs, and cannot be
 * built (no nvcc/Boost/Eigen; HEAD has compile errors) — "parity unpinned" for those stages;
 * they are argued by line-by-line correspondence (citations below), an independent float64
 * numpy restatement in tests/, and invariants.
 */
#ifndef SCPHD_CPU_H
#define SCPHD_CPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/slamtypes.h:123-127 */
typedef struct { float cov[4]; float mean[2]; float weight; } o_gaussian;
/* src/slamtypes.h:44-51 */
typedef struct { float px, py, ptheta, vx, vy, vtheta; } o_pose;
/* src/slamtypes.h:96-101 */
typedef struct { float range, bearing; int32_t label; } o_meas;

/* the SlamConfig fields that reach the hot path */
typedef struct {
    float dt;
    float minRange, maxRange, maxBearing;
    float stdRange, stdBearing;
    float clutterDensity;
    float pd;
    float birthWeight, birthNoiseFactor;
    float minFeatureWeight, minSeparation;
    float resampleThresh;
    float l, h, a, b;
    int32_t subdividePredict;
    int32_t distanceMetric;        /* 0 Mahalanobis, 1 Hellinger */
    int32_t labeledMeasurements;
    int32_t particleWeighting;     /* only 0 supported */
    /* how o_merge adds a cluster's members (the reference uses a block-size-dependent reduction tree,
     * src/phdfilter.cu:2795-2881: no order of its own):
     *   0  exact, order-free integer sums of the reference's terms (what the device computes; see o_merge)
     *   1  float sums in (weight desc) order, seed first — the order of src/gm_reduce.cpp:103-118 */
    int32_t mergeSums;
} o_config;

/* per-thread scratch stack of the per-particle routines (scphd_cpu.c: why) */
typedef struct { int block; size_t top; } o_tmp_frame;
o_tmp_frame o_tmp_enter(void);
void* o_tmp_alloc(size_t bytes);
void o_tmp_leave(o_tmp_frame f);
void o_tmp_release(void);
int o_omp_max_threads(void);

float o_safe_log(float x);
float o_wrap_angle(float a);
double o_det_exp(float x);

/* src/phdfilter.cu:797-823; noise[i] = {n_alpha, n_encoder} */
void o_predict_ackerman(o_pose* poses, int n, float alpha, float v_encoder,
                        const float* noise, const o_config* cfg);

/* src/phdfilter.cu:1328-1332,1841-1845 */
void o_predicted_measurement(const o_pose* pose, const float* mean, float* r_out, float* r2_out,
                             float* b_out, float* dx_out, float* dy_out);

/* src/phdfilter.cu:1328-1346; cls: 1 in range, 2 nearly in range, 0 out */
void o_classify(const o_gaussian* map, int n, const o_pose* pose, const o_config* cfg, int8_t* cls);

/* src/phdfilter.cu:3470-3506; weight = safeLog(birthWeight) */
void o_births(const o_pose* pose, const o_meas* z, int M, const o_config* cfg, o_gaussian* births);

/* src/phdfilter.cu:1833-1924 for the n in-range features of one particle:
 * pd[n]; preupdate[m*n + i] with LOG weight */
void o_preupdate(const o_pose* pose, const o_gaussian* feat, int n, const o_meas* z, int M,
                 const o_config* cfg, float* pd, o_gaussian* preupdate);

/* src/phdfilter.cu:2119-2319 for one particle: slab[n*(M+1)+M], prune flags, Δ log-weight */
void o_update(const o_gaussian* feat, const float* pd, const o_gaussian* preupdate,
              const o_gaussian* births, int n, int M, const o_config* cfg,
              o_gaussian* slab, uint8_t* prune_flag, float* dlogw);

/* src/device_math.cuh:308-325 / :373-413 */
float o_mahal_dist(const o_gaussian* a, const o_gaussian* b);
float o_hellinger_dist(const o_gaussian* a, const o_gaussian* b);

/* src/phdfilter.cu:2739-2890 semantics: greedy max-weight seed, merge d < minSeparation,
 * moment matching; tie-break lowest index; sums in (weight desc, index asc) order.
 * out must hold n entries; returns the merged count.  margin_out (optional, 2 floats):
 * [0] = min |d - minSeparation| / minSeparation over all distance decisions,
 * [1] = min relative weight gap between a seed and the next unmerged candidate. */
int o_merge(const o_gaussian* in, int n, const o_config* cfg, o_gaussian* out, float* margin_out);

/* literal transcription of src/gm_reduce.cpp:57-134 (sort, deque, Cholesky distance :30-37) */
int o_gm_reduce(const o_gaussian* in, int n, float min_distance, o_gaussian* out);

/* computeExpectedMap (src/main.cpp:290-316): weighted concatenation of all particle maps -> o_gm_reduce */
int o_expected_map(const o_gaussian* maps, const int32_t* sizes, const float* logw, int n_particles,
                   float min_distance, o_gaussian* out);

/* full per-particle measurement update: classify, births, pre-update, update, prune,
 * recombine with near-range, merge, append out-of-range (src/phdfilter.cu:3336-3761 for one
 * particle).  map_out must hold n_map*(M+1)+M+n_map entries.
 * survivors_out/surv_slab_idx (optional): pruned slab followed by the near-range features.
 * Returns the new map size. */
int o_update_particle(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                      const o_config* cfg, o_gaussian* map_out, float* dlogw,
                      o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out,
                      float* margin_out);

/* the same + (test diagnostic) slab_all_out: the whole unpruned slab followed by the nearly-in-range features — the array
 * surv_slab_idx indexes, n_in (M + 1) + M + n_near entries (pass a buffer of n_map (M + 2) + M) */
int o_update_particle_ex(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                         const o_config* cfg, o_gaussian* map_out, float* dlogw,
                         o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out,
                         float* margin_out, o_gaussian* slab_all_out);

/* TEST DIAGNOSTIC (no counterpart in the reference): the merge of `in` taking every decision from the same merge of `ref`
 * (the device's survivors), with a first-order proof for every decision `in` alone would have taken differently —
 * see the comment at the definition.  out: n entries; stats: 9 doubles.  Returns the merged count. */
int o_merge_follow(const o_gaussian* ref, const o_gaussian* in, int n, const o_config* cfg, o_gaussian* out, double* stats);

/* ---- CPHD variant (cphd_cpu.c; parity unpinned, see its header) ---- */
void o_cphd_log_factorials(float* lfact, int n);
/* 1: leave-one-out ESFs by M separate recursions (src/phdfilter.cu.bak:1247-1272, O(M^3)); 0 (default): O(M^2) */
void o_cphd_set_reference_esf(int on);
void o_cphd_terms(const float* cn_prior, int cn_len, const float* S, int M, float w_all, float pdw,
                  float birth_weight, float clutter_rate, float clutter_density,
                  float* lz, float* r1_out, float* cn_out, float* lY0_out);
void o_cphd_update(const o_gaussian* feat, const float* pd, const o_gaussian* preupdate, const o_gaussian* births,
                   int n, int M, const o_config* cfg, float clutter_rate, float w_all,
                   const float* cn_prior, int cn_len,
                   o_gaussian* slab, uint8_t* prune_flag, float* dlogw, float* cn_out, float* r1_out);
int o_cphd_update_particle(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                           const o_config* cfg, float clutter_rate, const float* cn_prior, int cn_len,
                           o_gaussian* map_out, float* dlogw, float* cn_out,
                           o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out, float* r1_out);

/* + margin_out[3] (merge distance margin, seed weight gap, prune margin) and the unpruned slab (test diagnostics) */
int o_cphd_update_particle_ex(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                              const o_config* cfg, float clutter_rate, const float* cn_prior, int cn_len,
                              o_gaussian* map_out, float* dlogw, float* cn_out,
                              o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out, float* r1_out,
                              float* margin_out, o_gaussian* slab_all_out);

int o_cphd_step(o_pose* poses, float* logw, o_gaussian* maps, int32_t* sizes, int n_particles, int cap,
                float alpha, float v_encoder, const float* noise, const o_meas* z, int M,
                const o_config* cfg, float clutter_rate, const float* cn, int cn_len, double uniform, int force_resample,
                o_gaussian* maps_out, int32_t* sizes_out, float* cn_out, int32_t* idx_out, float* neff_out, int n_threads);

/* src/phdfilter.cu:3741-3755 + src/device_math.cuh:549-558 */
void o_normalize_weights(float* logw, const float* dlogw, int n);
/* src/main.cpp:1281-1284 */
float o_neff(const float* logw, int n);
/* src/main.cpp:453-501 (n_uniforms == n_new: stratified) and src/phdfilter.cu.bak:3279-3327
 * (n_uniforms == 1: systematic) */
void o_resample(const float* logw, int n, const double* uniforms, int n_uniforms, int n_new, int32_t* idx);
/* src/main.cpp:331-361 */
void o_expected_pose(const o_pose* poses, const float* logw, int n, o_pose* out);
int o_argmax_weight(const float* logw, int n);

/* whole filter step over fixed-capacity slabs, OpenMP over particles (the timed CPU baseline):
 * maps[p*cap .. p*cap+sizes[p]); returns 0 or -1 on capacity overflow. */
int o_step(o_pose* poses, float* logw, o_gaussian* maps, int32_t* sizes, int n_particles, int cap,
           float alpha, float v_encoder, const float* noise, const o_meas* z, int M,
           const o_config* cfg, double uniform, int force_resample,
           o_gaussian* maps_out, int32_t* sizes_out, int32_t* idx_out, float* neff_out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
