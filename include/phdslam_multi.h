/*
 * phdslam_multi.h — ONE GM-PHD-SLAM filter sharded over the GPUs of one node, driven from C++ through a C-ABI
 * (libphdslam_multi.so = libphdslam.so + RCCL).
 *
 * The reference is single-GPU (src/main.cpp:1449 selects device 0).  SURVEY.md §8(b) proposed
 * `phd_create(const SlamConfig*, int n_devices, ...)`; this is that entry point: one process, one shard
 * (phd_filter, include/phdslam.h) and one HIP stream per device, an RCCL communicator over the devices
 * (ncclCommInitAll), and the per-step sequence of SURVEY.md §8(e) enqueued by ONE host thread:
 *
 *     every shard:  predict + update + prune + merge of its particles      (no communication)
 *     RCCL all-gather of the un-normalised log-weights                      (4 N bytes)
 *     every shard:  the identical normalise / nEff / resample-index routine (bit-identical indices)
 *     migration of the particles whose parent lives on another shard        (ncclSend/ncclRecv pairs, point to point over xGMI)
 *
 * Small shards (all shards' packed particles together below gathered_limit_bytes) exchange whole shards with ONE
 * in-place all-gather instead (the update kernel writes its rows into the shard's segment of the receive buffer) and never
 * wait for the host.  Results equal a single filter's bit for bit (the weights routine
 * is a pure function of the gathered vector; resampling moves whole slabs).
 *
 * Shards that share a device (more shards than GPUs: one-GPU test boxes) exchange by stream-ordered device copies
 * instead of RCCL — the same code path above the transport.
 */
#ifndef PHDSLAM_MULTI_H
#define PHDSLAM_MULTI_H

#include "phdslam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phd_multi phd_multi;

enum { PHD_TRANSPORT_AUTO = 0, PHD_TRANSPORT_RCCL = 1, PHD_TRANSPORT_PEER_COPY = 2 };
/* how a resampling step moves particles between shards:
 *   GATHERED  whole shards in one in-place all-gather (small shards; no host round trip)
 *   PULL      every shard reads its slots' parents straight out of the owners' slabs (peer access over xGMI; no host round
 *             trip, no staging buffer, a remote parent crosses the link once per destination shard)
 *   ALLTOALL  index download + host plan + export + ncclSend/ncclRecv pairs + import (one host round trip)
 *   AUTO      GATHERED while n_particles * pack bytes <= gathered_limit_bytes, else PULL when every device pair has peer
 *             access (always true for shards on one device), else ALLTOALL — never an error.  The environment variable
 *             PHD_MULTI_EXCHANGE = gathered | pull | alltoall | auto picks the form an AUTO create takes (an explicit option
 *             wins): the way back to the host-planned exchange on a machine whose peer reads misbehave.
 * The PULL form on DISTINCT devices is unmeasured on hardware (this build has only ever seen one-GPU boxes): its cross-device
 * ordering is the RCCL all-gather of the log-weights (every owner's update precedes it on the owner's stream; every reader's
 * pull follows it on the reader's stream) plus the `done` events the owners' next update waits for.  tests/test_gpu_multi_devices.py
 * runs it against a single filter bit for bit wherever two devices exist, and bench.py verifies it on first contact. */
enum { PHD_EXCHANGE_AUTO = 0, PHD_EXCHANGE_GATHERED = 1, PHD_EXCHANGE_ALLTOALL = 2, PHD_EXCHANGE_PULL = 3 };

/* zero-initialise and set what you need (0 = default) */
typedef struct {
    int32_t n_shards;             /* shards = ranks; 0 = the number of visible devices                                      */
    const int32_t* devices;       /* HIP ordinal per shard; NULL = shard k on device k mod (visible devices)                 */
    int32_t map_capacity;         /* as phd_options                                                                          */
    int32_t max_measurements;
    int32_t survivor_capacity;
    int32_t transport;            /* PHD_TRANSPORT_*: AUTO = RCCL when every shard has a device of its own, else peer copies */
    int32_t exchange;             /* PHD_EXCHANGE_*: AUTO = gathered while n_particles * pack bytes <= gathered_limit_bytes  */
    size_t gathered_limit_bytes;  /* 0 = 32 MiB                                                                              */
    uint32_t flags;               /* PHD_MULTI_FLAG_*                                                                        */
} phd_multi_options;
/* behave as if no pair of devices had peer access (tests the AUTO decision: it must fall back to ALLTOALL without an error) */
enum { PHD_MULTI_FLAG_NO_PEER_ACCESS = 1u };

/* cfg->n_particles = the GLOBAL particle count (divisible by n_shards).
 * replaces: the device setup of main() (src/main.cpp:1449-1466) + the per-step allocations of phdUpdateSynth
 * RCCL communicator creation runs under a watchdog (PHD_RCCL_INIT_TIMEOUT seconds, default 120, 0 = no limit).  On a time-out
 * the call returns PHD_ERR_HIP with the diagnostics to run, and THE PROCESS MUST EXIT: the bootstrap thread is still inside
 * ncclCommInitAll on these devices, so everything it may touch (the shards, their streams, NCCL_SOCKET_IFNAME) is deliberately
 * left alive, and a further phd_multi_create in the same process would race it.
 * NCCL_SOCKET_IFNAME: when unset, it is set to "lo" around ncclCommInitAll and removed afterwards (one process, one node: no
 * routable interface is needed) — setenv/unsetenv are not thread-safe against getenv in other threads, so either call this
 * before the process starts threads that read the environment, or export NCCL_SOCKET_IFNAME yourself (then nothing is touched). */
int phd_multi_create(const phd_slam_config* cfg, const phd_multi_options* opt, phd_multi** out);
int phd_multi_destroy(phd_multi* m);
int phd_multi_n_shards(const phd_multi* m);
int phd_multi_n_particles(const phd_multi* m);           /* global: n_particles of the configuration */
int phd_multi_n_particles_now(const phd_multi* m);       /* global, now: more between a shotgun predict (n_predict_particles > 1)
                                                            and the resample that follows (every shard grows alike) */
int phd_multi_uses_rccl(const phd_multi* m);             /* 1: RCCL collectives; 0: peer copies (shards share a device) */
int phd_multi_exchange_is_gathered(const phd_multi* m);  /* the form a forced resample takes */
int phd_multi_exchange(const phd_multi* m);              /* PHD_EXCHANGE_* in use (never AUTO) */
phd_filter* phd_multi_shard(phd_multi* m, int k);        /* shard k's filter (inspection, tests) */
int phd_multi_seed(phd_multi* m, uint64_t seed);
int phd_multi_set_config(phd_multi* m, const phd_slam_config* cfg);
int phd_multi_set_frozen(phd_multi* m, int freeze);
int phd_multi_sync(phd_multi* m);

/* SynthSLAM state of the GLOBAL particle set, in global particle order (shard k owns [k n, (k+1) n)) */
int phd_multi_set_particles(phd_multi* m, const phd_pose* poses, const float* log_weights, int n);
int phd_multi_get_particles(phd_multi* m, phd_pose* poses_out, float* log_weights_out);
int phd_multi_set_maps(phd_multi* m, const phd_gaussian2d* concat, const int32_t* sizes);
int phd_multi_get_map_sizes(phd_multi* m, int32_t* sizes_out);
int phd_multi_get_maps(phd_multi* m, phd_gaussian2d* concat_out, size_t concat_capacity, int32_t* sizes_out);

/* One filter step of run_synth's loop body (src/main.cpp:1244-1297) on the sharded filter:
 * phdPredict (noise: n_particles host entries in global order, or NULL = the device generator, which draws by GLOBAL
 * particle index) -> phdUpdateSynth(Z) -> global weight normalisation -> nEff -> resampleParticles when
 * nEff <= resample_threshold and the step had measurements (or always: force_resample).  n_meas == 0: predict only.
 * did_resample_out (optional) reports the decision.  Host synchronisations per step: none with the gathered exchange and
 * a forced resample; one (the nEff read) for the reference's trigger; one more (the index download, once for all shards)
 * when particles migrate by all-to-all. */
int phd_multi_step(phd_multi* m, phd_ackerman_control u, const phd_ackerman_noise* noise, const phd_measurement* z,
                   int n_meas, double uniform, int force_resample, int32_t* did_resample_out);
/* the same with the step's inputs already on the devices (bench: inputs resident in HBM) */
int phd_multi_upload_inputs(phd_multi* m, const phd_ackerman_noise* noise, const phd_measurement* z, int n_meas);
int phd_multi_step_resident(phd_multi* m, phd_ackerman_control u, double uniform, int force_resample,
                            int32_t* did_resample_out);

/* run_synth's order when the state is logged between the update and the resample (src/main.cpp:1260-1297):
 * phd_multi_update = phdPredict + phdUpdateSynth + the global weight normalisation (no resample; the global nEff is in the
 * step report of phd_multi_state_snapshot); phd_multi_resample = resampleParticles over the global set — the caller applies
 * the trigger (nEff <= resample_threshold and the step had measurements, :1286). */
int phd_multi_update(phd_multi* m, const phd_ackerman_control* u /* NULL: no motion (step 0, src/main.cpp:1244) */,
                     const phd_ackerman_noise* noise, const phd_measurement* z, int n_meas);
int phd_multi_resample(phd_multi* m, double uniform);

/* recoverSlamState (src/main.cpp:318-361) over the global set: weighted-mean pose (accumulated on the host in double,
 * in particle order), the map of the arg-max-weight particle (ties: lowest index), optionally all poses / log-weights,
 * and the step report (status words OR-ed over the shards, high-water marks max-ed, the global nEff). */
int phd_multi_state_snapshot(phd_multi* m, phd_pose* expected_out, phd_gaussian2d* map_out, int capacity,
                             int32_t* n_map_out, int32_t* particle_out, phd_pose* poses_out, float* log_weights_out,
                             phd_step_report* report_out);
/* computeExpectedMap (src/main.cpp:290-316) over the global set: every shard's weighted concatenation is copied to
 * shard 0, which reduces the global mixture (src/gm_reduce.cpp:57-134) on its device */
int phd_multi_expected_map(phd_multi* m, phd_gaussian2d* out, int capacity, int32_t* n_out);

/* Where a step's time goes ("report resample-with-migration time separately"): with timing enabled every
 * phd_multi_step_resident records HIP events on shard 0's stream at its phase boundaries and drains all shards at the end
 * (a separate pass: the timed loop runs without it).  us_total[PHD_MULTI_PHASES] accumulates microseconds per phase over
 * *steps_out steps: the local step (predict + update + prune + merge, one launch per shard), the RCCL all-gather, the
 * replicated weights / index routine, the index download + migration plan + export (the one host round trip of the
 * all-to-all form), the ncclSend/ncclRecv pairs, and copy_particles (import). */
enum { PHD_MULTI_PHASE_LOCAL_STEP = 0, PHD_MULTI_PHASE_ALL_GATHER = 1, PHD_MULTI_PHASE_WEIGHTS = 2,
       PHD_MULTI_PHASE_PLAN_EXPORT = 3, PHD_MULTI_PHASE_SEND_RECV = 4, PHD_MULTI_PHASE_IMPORT = 5, PHD_MULTI_PHASES = 6 };
int phd_multi_timing_enable(phd_multi* m, int enable);
int phd_multi_timing_reset(phd_multi* m);
int phd_multi_timing_read(phd_multi* m, double* us_total /*[PHD_MULTI_PHASES]*/, int64_t* steps_out);

#ifdef __cplusplus
}
#endif
#endif /* PHDSLAM_MULTI_H */
