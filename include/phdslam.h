/*
 * phdslam.h — C-ABI of the MI355X-native Rao-Blackwellised GM-PHD-SLAM hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI; its seam is
 * the C++ header src/phdfilter.h:10-34 plus the POD types of src/slamtypes.h.  Every
 * entry point below names the reference interface it replaces.  All types are POD with
 * the reference's exact memory layout (checked by static asserts in the implementation
 * and by tests/test_abi.py), all outputs are caller-allocated, all functions return an
 * int status (0 = PHD_OK, <0 = error; text via phd_last_error()).
 *
 * The library needs a gfx950 device for every compute call: there is no CPU fallback.
 * A compute call without a usable HIP device returns PHD_ERR_NO_DEVICE.
 */
#ifndef PHDSLAM_H
#define PHDSLAM_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * POD types — byte-for-byte the layouts of the reference's src/slamtypes.h
 * ---------------------------------------------------------------------------------- */

/* src/slamtypes.h:44-51 ConstantVelocityState: the vehicle pose for both motion models */
typedef struct {
    float px, py, ptheta, vx, vy, vtheta;
} phd_pose;

/* src/slamtypes.h:84-87 AckermanControl (alpha first, then v_encoder) */
typedef struct {
    float alpha;
    float v_encoder;
} phd_ackerman_control;

/* src/slamtypes.h:90-93 AckermanNoise */
typedef struct {
    float n_alpha;
    float n_encoder;
} phd_ackerman_noise;

/* src/slamtypes.h:96-101 RangeBearingMeasurement */
typedef struct {
    float range;
    float bearing;
    int32_t label;
} phd_measurement;

/* src/slamtypes.h:123-127 Gaussian2D: cov first, column-major 2x2, then mean, then weight */
typedef struct {
    float cov[4];
    float mean[2];
    float weight;
} phd_gaussian2d;

/* src/slamtypes.h:142-250 SlamConfig — same field order, same 324-byte layout.
 * (C has no bool: the reference's bool members are uint8_t here, same size/alignment.) */
typedef struct {
    uint8_t debug;
    float x0, y0, z0, roll0, pitch0, yaw0;
    float vx0, vy0, vz0, vroll0, vpitch0, vyaw0;
    uint8_t followTrajectory;
    float ax, ay, az, aroll, apitch, ayaw;
    float dt;
    float minRange, maxRange, maxBearing;
    float stdRange, stdBearing;
    float clutterRate, clutterDensity;
    float pd;
    float stdVxMap, stdVyMap;
    float stdAxMap, stdAyMap;
    float covVxBirth, covVyBirth;
    float ps;
    float tau, beta;
    int32_t particlesPerFeature, imageWidth, imageHeight;
    float stdU, stdV, disparityBirth, stdDBirth, fx, fy, u0, v0;
    int32_t n_particles;
    int32_t nPredictParticles;
    int32_t subdividePredict;
    float resampleThresh;
    float birthWeight;
    float birthNoiseFactor;
    uint8_t gateBirths;
    uint8_t gateMeasurements;
    float gateThreshold;
    float minExpectedFeatureWeight;
    float minSeparation;
    int32_t maxFeatures;
    float minFeatureWeight;
    int32_t particleWeighting;
    int32_t daughterMixtureType;
    int32_t nSamples;
    int32_t maxCardinality;
    int32_t filterType;
    int32_t distanceMetric;
    int32_t maxSteps;
    int32_t featureModel;
    int32_t motionType;
    int32_t mapEstimate;
    int32_t cphdDistType;
    float nu;
    uint8_t labeledMeasurements;
    float l, h, a, b;
    float stdAlpha, stdEncoder;
    uint8_t saveAllMaps;
    uint8_t savePrediction;
} phd_slam_config;

/* the reference caps the measurement set at 256 (src/phdfilter.cu:120,3390-3394) */
#define PHD_MAX_MEASUREMENTS 256

/* ------------------------------------------------------------------------------------
 * Status codes (reference: checkCudaErrors -> exit(EXIT_FAILURE); here: return codes)
 * ---------------------------------------------------------------------------------- */
enum {
    PHD_OK = 0,
    PHD_ERR_INVALID_ARG = -1,
    PHD_ERR_NO_DEVICE = -2,      /* no usable HIP device / HIP runtime error            */
    PHD_ERR_HIP = -3,
    PHD_ERR_UNSUPPORTED = -4,    /* config selects a branch SURVEY.md §2 marks out of scope */
    PHD_ERR_CAPACITY = -5,       /* a map or survivor list exceeded its configured capacity */
    PHD_ERR_NAN = -6,            /* NaN particle weights (reference: main.cpp:1307-1311)   */
    PHD_ERR_IO = -7,
    PHD_ERR_PARSE = -8
};

typedef struct phd_filter phd_filter; /* opaque: owns all device state of one rank's shard */

/* creation options; zero-initialise and set what you need (0 = default) */
typedef struct {
    int32_t n_particles;      /* particles held by THIS filter (a rank's shard); 0 = cfg->n_particles */
    int32_t map_capacity;     /* Gaussians per particle slab; 0 = 256                                   */
    int32_t max_measurements; /* <= PHD_MAX_MEASUREMENTS; 0 = 256: what a scan is clamped to (src/phdfilter.cu:3390-3394).  With map_capacity 512
                               * (128) and the default survivor capacity the filter RESERVES room for 64 (32) measurements — one of the LDS
                               * layouts the update kernel has compiled in (+3 % PHD ... +25 % CPHD on real scans); the clamp stays as given */
    int32_t survivor_capacity;/* pruned update components kept per particle before merging; 0 = auto (map_capacity + 8 max_measurements,
                               * at most 2048 — what LDS holds).  > 2048 adds a spill list in HBM: particles with more than 2048
                               * survivors (dense scans of large maps; the reference has no cap) are merged by a plain global-memory
                               * kernel — slower, same results — instead of failing with PHD_ERR_CAPACITY; at most 32768       */
    int32_t device;           /* HIP device ordinal                                                      */
    void*   stream;           /* hipStream_t to enqueue on; NULL = the filter creates its own            */
    int32_t global_particles; /* total particles over all ranks (for -log N after a global resample); 0 = n_particles */
    int32_t global_offset;    /* index of this shard's first particle in the global ordering            */
} phd_options;

const char* phd_last_error(void);
const char* phd_version(void);

/* ------------------------------------------------------------------------------------
 * Lifetime / configuration
 * ---------------------------------------------------------------------------------- */

/* replaces: device-state setup spread over phdUpdateSynth/prepareUpdateInputs
 * (src/phdfilter.cu:2966-3102,3403-3438: ~22 cudaMalloc per step) — one arena for the run */
int phd_create(const phd_slam_config* cfg, const phd_options* opt, phd_filter** out);
int phd_destroy(phd_filter* f);

/* replaces: setDeviceConfig(const SlamConfig&) (src/phdfilter.h:33-34, src/phdfilter.cu:3885-3890) */
int phd_set_config(phd_filter* f, const phd_slam_config* cfg);

/* replaces: initRandomNumberGenerators() (src/phdfilter.h:10) + rng.cpp's wall-clock seed;
 * seeds the counter-based generator used when predict is called without explicit noise */
int phd_seed(phd_filter* f, uint64_t seed);

/* ------------------------------------------------------------------------------------
 * State transfer (SynthSLAM <-> device; src/slamtypes.h:288-311)
 * ---------------------------------------------------------------------------------- */
int phd_n_particles(const phd_filter* f);
int phd_map_capacity(const phd_filter* f);

/* SynthSLAM::states / ::weights (log-weights) */
int phd_set_particles(phd_filter* f, const phd_pose* poses, const float* log_weights, int n);
int phd_get_particles(phd_filter* f, phd_pose* poses_out, float* log_weights_out);

/* SynthSLAM::maps_static — all maps at once: concat[sum sizes], sizes[n_particles] */
int phd_set_maps(phd_filter* f, const phd_gaussian2d* concat, const int32_t* sizes);
int phd_get_map_sizes(phd_filter* f, int32_t* sizes_out);
int phd_get_maps(phd_filter* f, phd_gaussian2d* concat_out, size_t concat_capacity, int32_t* sizes_out);
/* one particle's map */
int phd_set_map(phd_filter* f, int particle, const phd_gaussian2d* g, int n);
int phd_get_map(phd_filter* f, int particle, phd_gaussian2d* out, int capacity, int32_t* n_out);

/* ------------------------------------------------------------------------------------
 * The hot path
 * ---------------------------------------------------------------------------------- */

/* replaces: phdPredict(SynthSLAM&, AckermanControl) (src/phdfilter.h:16-17,
 * src/phdfilter.cu:1080-1257 + kernel :785-825).  noise: n_particles host entries drawn by the
 * caller in the reference's order (n_alpha, n_encoder); NULL = draw on device from phd_seed(). */
int phd_predict_ackerman(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* noise);

/* replaces: SynthSLAM phdUpdateSynth(SynthSLAM&, measurementSet) (src/phdfilter.h:23-24,
 * src/phdfilter.cu:3336-3761): in-range split, births, EKF pre-update, GM-PHD weight update,
 * prune, merge, particle log-weight increment and logSumExp normalisation.
 * n_meas > max_measurements is clamped like the reference (src/phdfilter.cu:3390-3394).
 * Asynchronous on the filter's stream. */
int phd_update(phd_filter* f, const phd_measurement* z, int n_meas);

/* phdPredict + phdUpdateSynth of ONE step in one launch (run_synth's src/main.cpp:1244-1272: the vehicle predict fused in front of
 * the update kernel, the weight normalisation as its tail): exactly the results of phd_predict_ackerman(f, u, noise) followed by
 * phd_update(f, z, n_meas) — which is what it runs where the fused launch does not apply (n_meas <= 0: predict only; the particle
 * shotgun; filters whose weights routine cannot be fused).  Asynchronous on the filter's stream. */
int phd_predict_update(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* noise, const phd_measurement* z, int n_meas);

/* replaces: the nEff test of run_synth (src/main.cpp:1281-1284) */
int phd_neff(phd_filter* f, float* neff_out);

/* replaces: resampleParticles + SynthSLAM::copy_particles (src/main.cpp:453-501,
 * src/slamtypes.h:313-333).  n_uniforms == 1: systematic (src/phdfilter.cu.bak:3279-3327);
 * n_uniforms == n_new: stratified (HEAD, src/main.cpp:468).  idx_out (optional) receives the
 * parent indices.  Weights become -log(n). Particle count is unchanged (n_new == n). */
int phd_resample(phd_filter* f, const double* uniforms, int n_uniforms, int32_t* idx_out);

/* run_synth's trigger (src/main.cpp:1286-1297): resample iff nEff <= resampleThresh (and the
 * step had measurements); did_resample_out reports the decision; idx_out as above (identity
 * when not resampled). */
int phd_resample_if_needed(phd_filter* f, double uniform, int had_measurements,
                           int32_t* did_resample_out, int32_t* idx_out);

/* replaces: recoverSlamState (src/main.cpp:318-361): weighted-mean pose and the map of the
 * arg-max-weight particle (MAP estimate). */
int phd_expected_pose(phd_filter* f, phd_pose* out);
int phd_map_estimate(phd_filter* f, phd_gaussian2d* out, int capacity, int32_t* n_out, int32_t* particle_out);

/* replaces: computeExpectedMap (src/main.cpp:290-316), the expected-a-posteriori map selected by
 * config.mapEstimate & 2 (src/main.cpp:363-379): every particle's map with feature weights scaled
 * by exp(particle log-weight), concatenated in particle order and reduced with
 * reduceGaussianMixture(concat, config.minSeparation) (src/gm_reduce.cpp:57-134) — on the device.
 * PHD_ERR_CAPACITY (with *n_out = size needed) when `capacity` is too small. */
int phd_expected_map(phd_filter* f, phd_gaussian2d* out, int capacity, int32_t* n_out);

/* replaces: reduceGaussianMixture<Gaussian2D> (src/gm_reduce.cpp:57-134) for any mixture, run on
 * the filter's device and stream.  Ties in weight are ordered by input index (the reference's
 * std::sort leaves them unspecified). */
int phd_gm_reduce(phd_filter* f, const phd_gaussian2d* in, int64_t n, float min_distance,
                  phd_gaussian2d* out, int capacity, int32_t* n_out);

/* CPHD variant (config filter_type = 1; configs[4] of BASELINE.json).  The reference's HEAD carries
 * the per-particle cardinality distributions (SynthSLAM::cardinalities, src/slamtypes.h:296, uniform
 * start src/main.cpp:1140-1143) but its CPHD kernels are commented out; the recursion implemented
 * here is the one of src/phdfilter.cu.bak:369-448,518-545,779-790,989-1503 (see DESIGN.md §CPHD for
 * the exact statement and the deviations).  phd_update/phd_step_dev run the CPHD update when the
 * filter was created with filter_type = 1; these calls expose the cardinality state:
 * rows of phd_cardinality_length() = max_cardinality + 1 log-probabilities, particle-major. */
int phd_cardinality_length(const phd_filter* f);
int phd_get_cardinalities(phd_filter* f, float* out);
int phd_set_cardinalities(phd_filter* f, const float* in);
/* replaces: cn_estimate of recoverSlamState (src/main.cpp:360): the arg-max-weight particle's row */
int phd_cardinality_estimate(phd_filter* f, float* out, int32_t* particle_out);

/* The two halves of phd_expected_map for the multi-GPU host: the weighted concatenation of this
 * rank's maps as SoA planes in device memory ([6][total]: weight, mean x, mean y, cov xx, xy, yy;
 * valid until the next call), and the reduction of `total` Gaussians given as n_planes = 6
 * (symmetric) or 7 (weight, mean x, mean y, cov[0], cov[1], cov[2], cov[3]) device planes. */
int phd_expected_map_concat_dev(phd_filter* f, float** d_planes, int64_t* total_out);
int phd_gm_reduce_dev(phd_filter* f, const float* d_planes, int64_t total, int n_planes, float min_distance,
                      phd_gaussian2d* out, int capacity, int32_t* n_out);
/* rounds (window/assign/compact passes) the last reduction took — diagnostics */
int phd_debug_gm_rounds(phd_filter* f);
/* global resamples of this shard that took the copy-free form (phd_global_resample_pull / _end: local parents by indirection,
 * remote ones parked in guest slabs; PHD_COPY_FREE=0 in the environment at phd_create selects the copying forms) — diagnostics */
int phd_debug_copy_free_resamples(phd_filter* f);
/* which instantiation of the update kernel the last update launch ran (csrc/phd_kernels.hip): below 18 the LDS layout and the scan's
 * length come from the arguments (any filter; PHD_LAYOUT=0 in the environment at phd_create keeps a filter there); 27 ... 35 the layout
 * of the filter is compiled in (map capacity 512 or 128 with the default survivor capacity), the scan's length is not - every real scan
 * on such a filter; 18 ... 26 both are compiled in - a full scan (as many measurements as the filter holds: the bench configurations).
 * Bit for bit the same results; diagnostics; -1 before any launch */
int phd_debug_update_instantiation(phd_filter* f);

/* ------------------------------------------------------------------------------------
 * Device-resident variants (inputs already in HBM; used by bench.py and the multi-GPU host)
 * ---------------------------------------------------------------------------------- */
int phd_predict_ackerman_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise);
int phd_update_dev(phd_filter* f, const phd_measurement* d_z, int n_meas);
/* device pointer to this shard's normalised log-weights (n_particles floats).  Valid only until the next hot-path call on
 * this handle: the weights routine writes out of place above 1024 particles and the library swaps its two buffers, so a
 * cached pointer may name the stale one — ask again after every step. */
int phd_logweights_dev(phd_filter* f, float** d_logw_out);
/* device pointer to the un-normalised log-weights after the update of the last phd_update
 * (before logSumExp); used for the multi-GPU all-gather */
int phd_raw_logweights_dev(phd_filter* f, float** d_logw_out);

/* Multi-GPU: update + prune + merge of the local shard WITHOUT the local normalisation: leaves
 * raw = logw + dlogw in the buffer of phd_raw_logweights_dev for the all-gather */
int phd_update_local_dev(phd_filter* f, const phd_measurement* d_z, int n_meas);
/* predict + local update in ONE launch (the vehicle predict and raw = logw + dlogw are done by the
 * update kernel); d_noise as in phd_predict_ackerman_dev */
int phd_step_local_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise,
                       const phd_measurement* d_z, int n_meas);

/* Multi-GPU: normalise / nEff / resample over the all-gathered vector of
 * n_global un-normalised log-weights (device pointer, identical on every rank).  Every rank
 * runs the identical routine -> identical indices.  Writes the shard's normalised weights
 * into the filter, returns nEff, and (if resample) the n_global parent indices (host). */
int phd_global_normalize(phd_filter* f, const float* d_all_logw, int n_global, float* neff_out);
int phd_global_resample_indices(phd_filter* f, const float* d_all_logw_normalized, int n_global,
                                const double* uniforms, int n_uniforms, int32_t* idx_out);
/* adopt parents for the local shard after a global resample: local_parent[i] >= 0 selects a
 * local particle; -1 means the slab arrives via phd_import_particle */
int phd_apply_parents(phd_filter* f, const int32_t* local_parent);
/* pack / unpack one particle (pose + map) for peer migration; buffer layout is private,
 * size = phd_particle_pack_bytes() */
size_t phd_particle_pack_bytes(const phd_filter* f);
int phd_export_particles_dev(phd_filter* f, const int32_t* particles, int n, void* d_buffer);
int phd_import_particles_dev(phd_filter* f, const int32_t* slots, int n, const void* d_buffer);
/* slot slots[k] <- row rows[k] of the buffer (rows == NULL: row k): a received parent may fill several slots */
int phd_import_particles_sel_dev(phd_filter* f, const int32_t* slots, const int32_t* rows, int n, const void* d_buffer);
int phd_finish_resample(phd_filter* f); /* weights <- -log(global_particles) */

/* The same exchange in two calls (one host round trip per resampling step).  begin: global indices
 * (phd_global_resample_indices on the gathered, normalised weights of phd_global_normalize), this
 * rank's part of the migration plan, export of the particles other ranks need into a library-owned
 * send buffer (grouped by destination rank; send_counts/recv_counts[world] in particles of
 * phd_particle_pack_bytes(); a parent travels once per destination rank however many of its slots it fills).  The caller runs all_to_all_single(recv, send, recv_counts,
 * send_counts) (RCCL).  end: copy_particles on the shard (local parents + received particles),
 * weights <- -log(global_particles).  idx_out (optional, host, global_particles entries).
 * d_all_raw_logw: NULL after phd_global_normalize; or the gathered UN-normalised weights, in which case
 * the normalisation and the indices come from one launch (forced resample, no nEff round trip). */
int phd_global_resample_begin(phd_filter* f, const float* d_all_raw_logw, double uniform, int world, int rank,
                              int32_t* send_counts, int32_t* recv_counts, void** d_send_buffer, int32_t* idx_out);
int phd_global_resample_end(phd_filter* f, const void* d_recv_buffer);
/* phd_global_resample_begin in two halves, for a host that drives several shards from one thread (libphdslam_multi.so):
 * every shard enqueues _launch (normalise — when d_all_raw_logw is given — and draw the global indices, no
 * synchronisation; *d_idx_out = the device copy of the indices); the host downloads the indices ONCE (they are identical on
 * every shard) and every shard plans its part of the migration and exports from that host copy with _plan. */
int phd_global_resample_launch(phd_filter* f, const float* d_all_raw_logw, double uniform, int32_t** d_idx_out);
/* the same from the gathered CURRENT (already normalised) log-weights of all shards — a resample no update precedes
 * (resampleParticles after a control-only step: the particle shotgun's N > 5 n_particles trigger, src/main.cpp:1286; a second
 * call of resampleParticles): the vector is used as it is, exactly as phd_resample uses a single filter's weights */
int phd_global_resample_launch_normalized(phd_filter* f, const float* d_all_logw, double uniform, int32_t** d_idx_out);
int phd_global_resample_plan(phd_filter* f, const int32_t* idx, int world, int rank, int32_t* send_counts,
                             int32_t* recv_counts, void** d_send_buffer);
/* The same exchange for small shards with no host round trip ("gathered" exchange): phd_export_shard_dev packs the
 * whole shard (n rows of phd_particle_pack_bytes; header word 7 = the particle's un-normalised log-weight) into the
 * library's send buffer; the caller all-gathers the shards in rank order (one fixed-size RCCL all-gather);
 * phd_global_resample_gathered takes the weights (weights_in_rows = 1: un-normalised, from the rows — normalised first;
 * 0: the vector phd_global_normalize left; 2: the shards' CURRENT normalised weights from the rows of
 * phd_export_shard_current_dev, used as they are — a resample no update precedes), draws the identical global indices on
 * every rank and fills this shard's slots straight from the gathered rows.  Everything is stream-ordered; idx_out (optional, host) forces a synchronisation.  Traffic is
 * world * n * pack bytes per rank: for small shards only — phd_global_resample_begin/_end move just the migrants. */
int phd_export_shard_dev(phd_filter* f, void** d_rows, size_t* bytes_out);
int phd_export_shard_current_dev(phd_filter* f, void** d_rows, size_t* bytes_out);   /* header word 7 = the current log-weight */
/* phd_step_local_dev + phd_export_shard_dev in ONE launch: the update kernel writes every particle's merged map, predicted
 * pose, count and raw log-weight straight into its export row.  The updated maps then exist only in the rows, so the step
 * MUST be completed by phd_global_resample_gathered; until then a live (not frozen) filter refuses every other call
 * except phd_sync / phd_device_status / phd_timing_* / phd_destroy. */
int phd_step_local_rows_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise,
                            const phd_measurement* d_z, int n_meas, void** d_rows, size_t* bytes_out);
/* In-place all-gather: d_rows != NULL makes phd_step_local_rows_dev write its rows there (n * phd_particle_pack_bytes bytes,
 * device; normally this rank's segment of the all-gather's receive buffer, so the collective skips the self copy) instead of
 * the library's send buffer; NULL restores the default.  The caller owns the buffer. */
int phd_set_rows_target(phd_filter* f, void* d_rows);
int phd_global_resample_gathered(phd_filter* f, const void* d_all_rows, double uniform, int world, int rank,
                                 int weights_in_rows, int32_t* idx_out);
/* The same exchange with NO host round trip and no staging buffer, for shards whose memory the devices can read directly
 * (one process: peer access over xGMI, or several shards on one device): after phd_global_resample_launch every shard holds
 * the identical global indices ON THE DEVICE; phd_global_resample_pull enqueues copy_particles (src/slamtypes.h:313-333) for
 * this shard's slots straight out of the owners' slabs — slot j <- particle idx[rank n + j] of shard idx / n, read through
 * views[owner] (phd_peer_view_get of every shard, taken AFTER their update and BEFORE this call).  A remote parent crosses the
 * link once per destination shard: the first of the (consecutive) slots it fills pulls it, the others copy from that slot.
 * The caller orders the streams: every owner's update must be complete before the pull starts (the log-weight all-gather
 * is such a point), and no owner may start its next update before every reader's pull is complete.
 * Replaces phd_global_resample_plan + the all-to-all + phd_global_resample_end. */
typedef struct {
    const float* maps;        /* [n][6][cap] slabs holding the updated maps          */
    const int32_t* counts;    /* [n]                                                 */
    const int32_t* parent;    /* [n] slab indirection, or NULL (identity)            */
    const phd_pose* poses;    /* [n] poses after the step's predict                  */
    const float* cn;          /* CPHD: [n][cn_len] cardinality rows (same indirection), else NULL */
} phd_peer_view;
int phd_peer_view_get(phd_filter* f, phd_peer_view* out);
int phd_global_resample_pull(phd_filter* f, const phd_peer_view* views, int world, int rank);
/* Round 5: both forms of the global resample (pull, and _plan/_end) move no LOCAL map while the shard's indirection is the
 * identity (no resample since its last update): a local parent stays where it is — the slot's indirection names its slab, as
 * in a single filter — and a remote parent is copied once, by the first slot it fills, into that slot's GUEST slab (the map
 * buffers of a shard are one allocation [buffer 0 | buffer 1 | guests]; the indirection counts slabs from the current
 * buffer).  PHD_COPY_FREE=0 in the environment at phd_create keeps the copying forms.
 *
 * The nEff-triggered step with no host round trip (the reference reads nEff back, src/main.cpp:1281-1296):
 * phd_global_resample_launch_auto normalises the gathered un-normalised weights, takes nEff and the decision ON THE DEVICE
 * and leaves the resampling indices or the identity; phd_global_resample_pull_auto — enqueued whatever the decision was, and
 * always after the former — completes the step either way (no resample: every slot keeps its particle and adopts its slice of
 * the normalised weights).  nEff and the decision are in the step report (phd_step_report_get) for a host that wants them.  Available while
 * phd_global_resample_auto_supported says 1 (the copy-free form applies); otherwise phd_global_normalize + the host's decision. */
/* the buffer phd_step_local_dev / phd_update_local_dev leave the un-normalised log-weights in (phd_raw_logweights_dev returns it): the
 * caller's (n_max floats) instead of the filter's own — e.g. this shard's segment of the all-gather's receive buffer, which makes the
 * collective in place.  NULL: the filter's own again. */
int phd_set_raw_target(phd_filter* f, float* d_raw);
int phd_global_resample_auto_supported(phd_filter* f, int world);
int phd_global_resample_launch_auto(phd_filter* f, const float* d_all_raw_logw, double uniform);
int phd_global_resample_pull_auto(phd_filter* f, const phd_peer_view* views, int world, int rank);

/* ------------------------------------------------------------------------------------
 * Bench / steady-state protocol and instrumentation
 * ---------------------------------------------------------------------------------- */
/* freeze != 0: steps read the current state but do not commit their outputs, so every
 * iteration restarts from the same device-resident snapshot (constant work per iteration) */
int phd_set_frozen(phd_filter* f, int freeze);
/* one full filter step on device-resident inputs: predict -> update -> prune -> merge ->
 * weight normalise -> nEff -> resample (forced if force_resample) */
int phd_step_dev(phd_filter* f, phd_ackerman_control u, const phd_ackerman_noise* d_noise,
                 const phd_measurement* d_z, int n_meas, double uniform, int force_resample);
int phd_sync(phd_filter* f);
void* phd_stream(phd_filter* f);

/* per-kernel timing with HIP events on the filter's stream (reference: cudaEvent pair in
 * phdPredict, src/phdfilter.cu:1083-1087,1244-1251).  enable, run steps, then read the
 * accumulated milliseconds and launch counts. */
enum { PHD_K_PREDICT = 0, PHD_K_UPDATE_MERGE = 1, PHD_K_WEIGHTS = 2, PHD_K_COUNT = 3 };
int phd_timing_enable(phd_filter* f, int enable);
int phd_timing_read(phd_filter* f, double* ms_total /*[PHD_K_COUNT]*/, int64_t* launches /*[PHD_K_COUNT]*/);
int phd_timing_reset(phd_filter* f);

/* ------------------------------------------------------------------------------------
 * Inspection (parity tests): the pruned update components of one particle before merging,
 * sorted into the reference's slab order [non-detect | detect m-major | births | near-range]
 * (src/phdfilter.cu:2123-2172,3218-3257); slab_index_out (optional) = position in the
 * un-pruned slab.  Valid after phd_update until the next hot-path call.
 * Also the per-particle log-weight increments of the last update (src/phdfilter.cu:2260-2263).
 * ---------------------------------------------------------------------------------- */
int phd_debug_enable(phd_filter* f, int enable); /* bit 0: survivor inspection, bit 1: phase stamps, bit 2: staged launches only (no fused weights tail) */
/* phase stamps of the last update (diagnostic kernel instantiation): out[n_particles][32] (+ 8 of the weights routine), 100 MHz ticks */
int phd_debug_get_stamps(phd_filter* f, uint64_t* out);
int phd_debug_get_survivors(phd_filter* f, int particle, phd_gaussian2d* out, int32_t* slab_index_out,
                            int capacity, int32_t* n_out);
int phd_debug_get_weight_increments(phd_filter* f, float* dlogw_out);
/* how the update kernel of this filter sits on a CU: dynamic LDS bytes per workgroup (one workgroup = one particle) and the
 * workgroups the runtime says a CU holds at a time (registers, LDS and waves taken together; 3 at 4096 x 256 x 64) - for the build
 * of the kernel this filter was given at phd_create (fixed for its life; environment PHD_UPDATE_BUILD=2|3 forces one).  Diagnostic. */
int phd_update_residency(phd_filter* f, int32_t* workgroups_per_cu_out, uint64_t* lds_bytes_out);
/* status word accumulated on the device: bit0 map overflow, bit1 survivor overflow (-> PHD_ERR_CAPACITY), bit2 the weights
 * workgroup(s) of a fused step - or a workgroup of the block-form weights routine (more than 4096 particles) at one of its two
 * grid-wide barriers - gave up waiting (-> PHD_ERR_HIP; a bounded spin of seconds, never seen in practice; the error text names
 * the wait and what it read) */
int phd_device_status(phd_filter* f, uint32_t* status_out, int32_t* max_survivors_out, int32_t* max_map_out);
/* everything the driver loop tests after a step, in ONE download (replaces the separate phd_device_status + phd_neff
 * round trips; run_synth's tests: src/main.cpp:1281-1311): the sticky status word and high-water marks, and the nEff /
 * resample decision left by the step's weights routine (phd_update, phd_resample_if_needed, phd_step_dev: nEff of the
 * normalised weights BEFORE any resample, as the reference's NaN exit needs it).  Returns what phd_device_status returns;
 * after a fused-step time-out (status bit 2) the hand-off counter is reset so that later steps are sound. */
typedef struct {
    uint32_t status;
    int32_t max_survivors, max_map;
    float neff;
    int32_t did_resample;
} phd_step_report;
int phd_step_report_get(phd_filter* f, phd_step_report* out);
/* the host-side mirror of a SynthSLAM whose particle count changed outside the library (phdPredict grows it by
 * n_predict_particles, src/phdfilter.cu:1185-1238; resampleParticles shrinks it, src/main.cpp:1289): the filter now holds
 * n particles (1 <= n <= n_particles, or 5 n_particles n_predict_particles with the shotgun); upload particles and maps next */
int phd_set_particle_count(phd_filter* f, int n);
/* recoverSlamState (src/main.cpp:318-361) in one call and one host synchronisation: weighted-mean pose, the map of the
 * arg-max particle (map_out[capacity], *n_map_out entries, *particle_out its index), and optionally every particle's pose and
 * log-weight — what phd_expected_pose + phd_map_estimate + phd_get_particles return in three round trips */
int phd_state_snapshot(phd_filter* f, phd_pose* expected_out, phd_gaussian2d* map_out, int capacity, int32_t* n_map_out,
                       int32_t* particle_out, phd_pose* poses_out, float* log_weights_out, phd_step_report* report_out /* optional */);

/* The same snapshot WITHOUT the host synchronisation, for a driver loop that keeps the device busy (replaces the blocking part of
 * recoverSlamState + the nEff read-back of run_synth, src/main.cpp:1274-1297; the reference logs between the update and the
 * resample - that ORDER of contents is kept, the host's wait is not).  Two slots:
 *   phd_snapshot_capture(f, slot)   after the step's update, BEFORE its resample: enqueues the state extraction and one pack of
 *                                   (expected pose, arg-max particle's map, every pose and log-weight) into the slot's device block;
 *   phd_snapshot_send(f, slot, idx) after the resample was enqueued: adds the step report (nEff of the pre-resample weights, the
 *                                   resample decision; idx != 0: the resample's parent indices too) and starts ONE download on a
 *                                   second stream - the filter's stream is free for the next step;
 *   phd_snapshot_wait(f, slot, &v)  blocks until that download is complete; v points INTO the slot's pinned block, valid until the
 *                                   slot is captured again.  Returns what phd_state_snapshot returns (PHD_ERR_NAN, PHD_ERR_CAPACITY ...).
 * phd_host_alloc / phd_host_free: page-locked host memory, so that phd_predict_ackerman's noise upload is asynchronous too. */
typedef struct {
    const phd_pose* expected;
    const phd_gaussian2d* map;
    const phd_pose* poses;
    const float* log_weights;
    const int32_t* resample_idx;   /* NULL unless asked for in phd_snapshot_send */
    const float* cardinality;      /* CPHD: cn_estimate, the arg-max particle's log cardinality row (src/main.cpp:360); NULL for a PHD filter */
    int32_t n_map, particle, n_particles, cardinality_len;
    phd_step_report report;
} phd_snapshot_view;
int phd_snapshot_capture(phd_filter* f, int slot);
int phd_snapshot_send(phd_filter* f, int slot, int want_resample_idx);
int phd_snapshot_wait(phd_filter* f, int slot, phd_snapshot_view* out);
void* phd_host_alloc(size_t bytes);
void phd_host_free(void* p);

/* ------------------------------------------------------------------------------------
 * Host-side boundary helpers (no device needed): config file, data files, log writer
 * ---------------------------------------------------------------------------------- */
/* replaces: loadConfig (src/main.cpp:956-1073): same keys, same defaults, derives
 * clutterDensity (src/main.cpp:1065-1066).  data_dir_out/n_steps_out receive the two
 * non-SlamConfig keys.  Unknown keys and malformed values are errors (reference ignores them). */
int phd_config_defaults(phd_slam_config* cfg);
int phd_config_load(const char* path, phd_slam_config* cfg, char* data_dir_out, size_t data_dir_cap,
                    int32_t* n_steps_out);

/* replaces: loadMeasurements/parseMeasurements (src/main.cpp:192-240): header line skipped,
 * one step per line, pairs "r b" (README:21-24) or triples "r b label" (HEAD parser).
 * Two-call protocol: first with out == NULL to get counts. */
int phd_load_measurements(const char* path, int triples, phd_measurement* out, size_t out_capacity,
                          int32_t* step_sizes_out, size_t steps_capacity, size_t* n_steps_out, size_t* n_total_out);
/* replaces: loadControls (src/main.cpp:169-190): header skipped, "v_encoder alpha" per line,
 * ',' tolerated as separator */
int phd_load_controls(const char* path, int has_header, phd_ackerman_control* out, size_t capacity, size_t* n_out);

/* replaces: loadTimestamps (src/main.cpp:147-166): one value per line, no header; a missing file
 * yields zero timestamps (= lock-step mode) */
int phd_load_timestamps(const char* path, float* out, size_t capacity, size_t* n_out);
/* replaces: loadTrajectory (src/main.cpp:242-260): "px py ptheta vx vy vtheta" per line, '%' lines skipped */
int phd_load_trajectory(const char* path, phd_pose* out, size_t capacity, size_t* n_out);

/* replaces: HEAD's 7-line writeLog (src/main.cpp:848-954): pose / static map / dynamic map / weights /
 * poses / resample indices / cardinality, append mode, weights and poses repeated
 * n_predict_particles times at step 0 */
int phd_write_state_log7(const char* dir, int step, const phd_pose* expected_pose, const phd_gaussian2d* map,
                         int n_map, const float* log_weights, const phd_pose* poses, const int32_t* resample_idx,
                         int n_particles, int max_cardinality, int n_predict_particles);

/* the same two writers with filter_type = CPHD: the cardinality line carries cn_estimate[0..cn_len)
 * instead of zeros (src/main.cpp:944-949) */
int phd_write_state_log_cphd(const char* dir, int step, const phd_pose* expected_pose, const phd_gaussian2d* map,
                             int n_map, const float* log_weights, const phd_pose* poses, int n_particles,
                             const float* cn_estimate, int cn_len);
int phd_write_state_log7_cphd(const char* dir, int step, const phd_pose* expected_pose, const phd_gaussian2d* map,
                              int n_map, const float* log_weights, const phd_pose* poses, const int32_t* resample_idx,
                              int n_particles, int n_predict_particles, const float* cn_estimate, int cn_len);

/* replaces: the state_estimate%05d.log contract (README:31-39; writer src/main.cpp:848-954):
 * 5 lines: pose / map (weight mx my c0 c1 c2 c3) / log-weights / poses / cardinality zeros */
int phd_write_state_log(const char* dir, int step, const phd_pose* expected_pose,
                        const phd_gaussian2d* map, int n_map, const float* log_weights,
                        const phd_pose* poses, int n_particles, int max_cardinality);

/* ------------------------------------------------------------------------------------
 * Estimation-quality metrics of the reference's offline tooling (host side, no device)
 * ---------------------------------------------------------------------------------- */
/* replaces: ospa_distance(X, Y, p, c) (python/ospa.py:220-274; its Munkres cannot be built here):
 * X[m][2], Y[n][2]; out[3] = (OSPA, localisation part, cardinality part) */
int phd_ospa(const float* X, int m, const float* Y, int n, double p, double c, double* out);
/* replaces: compute_error_k (python/batch_analyze.py:16-37) on one state_estimate log:
 * out[5] = (pose error, OSPA, OSPA localisation, OSPA cardinality, nEff); the map estimate is the
 * round(sum of weights) highest-weighted features; the reference uses p = 1, c = 5 */
int phd_evaluate_state_log(const char* path, const float* true_pose_xy, const float* true_map, int n_true,
                           double p, double c, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PHDSLAM_H */
