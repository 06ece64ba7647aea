// phd_kernels.hip — gfx950 (CDNA4) kernels of the GM-PHD-SLAM hot path: the fused update + prune + merge kernel, the
// predict and utility kernels and the host-side launchers.  One translation unit; the device code it is made of:
//
//   phd_defs.h     workgroup shape (PHD_NW = 8 waves, 512 threads), typedefs, the LDS pointer qualifier
//   phd_lane.h     cross-lane exchange (DPP / v_permlane*_swap), wave and workgroup reductions and scans
//   phd_math.h     EKF terms per feature, Joseph covariance, Mahalanobis / Hellinger distance
//   phd_lds.h      LDS pointers of the update kernel, counters, survivor slot allocation (byte layout: phd_lds_layout.h — sized
//                  for THREE workgroups per CU at 4096 x 256 x 64)
//   phd_sort.h     the merge's first sort: counting sort on the weight key, rank by counting, register bitonic network
//   phd_merge.h    exact closeness decision, merge_small (<= 256 survivors, one shot), merge rounds, grouping, moments
//   phd_predict.h  Ackerman predict, counter-based noise generator
//   phd_cphd.h     the CPHD block (cardinality, ESF sweeps)
//   phd_weights.h  weights / nEff / fixed-point-CDF resample routine (stand-alone kernels and the fused tail)
//
// One workgroup (512 threads = 8 wave64) owns one particle for the whole measurement update:
//
//   classify map (in range / nearly in range / out)        reference: computeInRangeKernel  src/phdfilter.cu:1279-1358
//   per-feature EKF terms -> LDS (SoA)                      reference: preUpdateSynthKernel  src/phdfilter.cu:1824-1925
//   pass 1: per-measurement normalisers  Z_m (+ the list of terms that can survive the prune)   phdUpdateKernel :2190-2223
//   pass 2: final weights of the listed terms, prune BEFORE store, survivors -> LDS   :2226-2245,2307-2319 + pruneMap :3120-3174
//   greedy merge of the survivors, entirely in LDS          reference: phdUpdateMergeKernel  src/phdfilter.cu:2707-2898
//   merged map + untouched out-of-range features -> HBM     reference: mergeAndCopyMaps      src/phdfilter.cu:3304-3318
//   (fused step) the last workgroup to finish runs the weights / nEff / resample routine      :3735-3755, main.cpp:453-501
//
// The reference materialises all G(M+1)+M update components per particle in global memory
// twice (1.9 GB at 4096x256x64) and makes 3 global passes per merged Gaussian.  Here the update
// components never leave registers unless they survive pruning, survivors live in LDS, and HBM
// sees only the map read, the map write and a few bytes per particle.
//
// Lane mapping of pass 1: lanes <-> features, a wave's chunk of 8 measurements in registers (phd_pass1.h), so the
// per-measurement sums over features are private accumulators — no LDS traffic and no cross-lane step in the inner
// loop (one transposing reduction per chunk at the end).  Pass 2 visits the listed terms one per thread.
//
// Residency: the kernel is latency-bound per workgroup, so what a CU holds at a time is most of the throughput.  Two builds of
// the template (launch bound MINW): 4 waves per SIMD = two workgroups per CU, 6 = three (80 VGPRs, a small scratch frame that
// stays out of the hot loops); launch_update_merge picks by the filter's LDS bytes and the launch's size.
//
// Merge: exact greedy semantics (seed = max weight, ties -> lowest slab index; absorb d < T); see phd_merge.h.
// A conservative trace bound (d >= 2|dm|^2/(tr Pa + tr Pb)) rejects far pairs before the exact
// 2x2 inverse; it never rejects a pair the exact test would accept (1 % guard band).
//
// No CUDA compatibility layer, no Thrust/hipCUB, wave64 only.

#include <mutex>
#include <set>
#include <map>
#include <atomic>

#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"
#include "phd_pass1.h"
#include "phd_sort.h"
#include "phd_merge.h"
#include "phd_predict.h"
#include "phd_cphd.h"
#include "phd_weights.h"
// (partial translation units: -DPHD_CPHD_TU / -DPHD_W6_TU / -DPHD_CPHD_W6_TU compile the kernel template and one table of
//  instantiations each)
#if defined(PHD_CPHD_TU) || defined(PHD_W6_TU) || defined(PHD_CPHD_W6_TU)
#define PHD_PART_TU 1
#endif
#ifndef PHD_PART_TU
#include "phd_spill.h"      // (phd_merge_spill_kernel: a plain __global__ function, defined once)
#endif

namespace phd {

// host-visible sizes (declared in phd_device.h)
// (PHD_LDS_PAD: occupancy experiments — extra bytes per workgroup; -DPHD_LDS_PAD=16384 leaves ONE workgroup per CU at the
// headline size: 578 us per launch against 348 with two, per-workgroup time 36.6 against 43.2 us — DESIGN.md §9)
#ifndef PHD_LDS_PAD
#define PHD_LDS_PAD 0
#endif
#ifndef PHD_PART_TU
size_t update_lds_bytes(int S, int C, int MM) { return lds_offsets(S, C, MM).total + PHD_LDS_PAD; }
// the fused step: up to 4096 particles weights_body<PHD_T, 8> in one workgroup of the launch, above (up to 2^18) the block form
// on grid_weights_workgroups(n) workgroups behind the particles' (phd_weights.h)
int update_fuse_max_particles() { return 1 << 18; }
int weights_grid_min_particles() { return PHD_GRID_WEIGHTS_MIN; }
int weights_grid_workgroups(int n) { return grid_weights_workgroups(n); }
size_t weights_grid_lds_bytes(int n) { return grid_weights_lds_bytes(n); }

// the block form of the weights routine as a launch of its own (a plain __global__ function: defined once, here)
__global__ __launch_bounds__(PHD_T) void phd_weights_grid_kernel(WeightArgs A, unsigned* status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_gwdyn[];
    // An EARLIER launch of this filter gave up at one of its waits (status bit, sticky until the host reads the step report and
    // re-zeroes the counters, phd_api.cpp: interpret_report): the barrier counters are mid-count, and a launch enqueued before the
    // host has looked would pass its barriers early and read incomplete block records.  It refuses instead: the bit
    // stays set, the host's next report fails the step loudly, and nothing downstream consumes half-made weights as good ones.
    if (status && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PHD_STATUS_TAIL_TIMEOUT)) return;
    weights_grid_body<false>(A, (int)blockIdx.x, (int)gridDim.x, s_gwdyn, status, nullptr);
}
size_t cphd_lds_bytes(int S_cap, int cn_len, int MM) { return cphd_extra_lds_bytes(S_cap, cn_len, MM); }

#endif // !PHD_PART_TU

// the last step of a particle's hand-off to the weights workgroup of the fused step: its hand-off stores (pose, indirection
// reset, log-weight increment — relaxed agent-scope atomics by the same thread, i.e. sc1 write-through stores) must be
// complete before the ticket counts the particle.  Two forms, A/B-measured (profiles/r03_ab_handoff.txt, tools/ab_bench.sh):
//   default             drain this thread's stores (s_waitcnt vmcnt(0)), then a relaxed agent-scope increment
//   -DPHD_HANDOFF_MM    the C++ memory-model form: a RELEASE increment here, an ACQUIRE fence after the consumer's poll.
//                       The release at agent scope also writes back the XCD's L2 (buffer_wbl2 sc1) — needed for plain stores,
//                       redundant for these write-through ones — which costs 1.4 us on the one-workgroup-per-CU critical path:
//                       256 x 64 x 32 runs 53.3 k instead of 57.5 k steps/s (-7.9 %); 4096 x 256 x 64 is unchanged (2842 vs 2845).
// The default therefore stays the drained form; what it relies on is that the compiler keeps the atomic stores, the asm
// statement ("memory" clobber) and the atomic increment in program order — all three are volatile-like side effects it may
// not reorder — and that sc1 stores are visible at agent scope once vmcnt has drained (cdna guide, Guideline 16).
__device__ __forceinline__ void handoff_ticket(unsigned* ticket)
{
#ifdef PHD_HANDOFF_MM
    __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// ------------------------------------------------------------------------------------------
// the fused update + prune + merge kernel
// ------------------------------------------------------------------------------------------
// SPILL: the instantiation of filters created with a spill list (survivor_capacity > 2048); without it the spill branches
// fold away (they cost 0.3 us of the 17.5 us step at 256 x 64 x 32 when merely present)
// MINW: waves per SIMD the register allocation must allow.  4 = two workgroups per CU (<= 128 VGPRs, what the code wants: 107);
// 6 = three (<= 80 VGPRs: ~50 registers live in scratch — measured 1.7 % slower at equal residency, 22 % faster when LDS lets the
// third workgroup in, profiles/r04_three_workgroups.txt).  launch_update_merge picks by the filter's LDS size.
// GRIDT: the fused step's tail is the BLOCK FORM of the weights routine on several workgroups (launches of more than 4096 particles,
// grid N + W; phd_weights.h) - an instantiation of its own, so that every other one keeps, textually, the code it had: the routine
// inlined beside the single tail workgroup cost the 80-register CPHD build 3.3 % (27 more spilled scalars from a changed branch
// condition alone) and raised the headline build's spills from 451 to 537 VGPRs.
// LAYOUT: the filter's LDS layout (survivor capacity, map capacity, measurement capacity) as COMPILE-TIME constants (round 5): with
// the layout known every LDS array is an immediate offset in its ds_* instruction instead of one of ~35 scalar registers — the
// kernel keeps ~100 scalars alive and spills the excess into vector-register lanes (v_writelane / v_readlane: vector instructions)
// — and an index shifted once serves all planes.  0 = from the arguments (any filter); 1 = BASELINE.json's 256-Gaussian
// configurations (configs[2], [3], [4]: 1024 / 512 / 64), 2 = configs[1] (512 / 128 / 32).  Same library, same visit, PHD_LAYOUT=0
// against the default: 4096 x 256 x 64 4 260 -> 4 270 steps/s, CPHD 2 714 -> 2 816 (+3.8 %), 16 384 x 256 x 64 1 232 -> 1 285
// (+4.3 %), 256 x 64 x 32 61.7 k -> 62.9 k (+1.9 %).  (A third layout, for bench.py's dense-scan rider - 2048 / 768 / 256 with a
// spill list - measured nothing, 150.0 against 150.2 steps/s, and was not kept.)  These instantiations also take the MEASUREMENT
// COUNT as a constant — a full scan, M = the measurement capacity, which is what those configurations are: on top of the layout,
// same visit, 4 342 -> 4 468 steps/s at the headline (+2.9 %), CPHD 2 808 -> 3 001 (+6.9 %: the block's tile count and sweep lengths
// fold), 16 384 particles 1 283 -> 1 328, 256 x 64 x 32 64.1 k -> 66.2 k.  The launcher picks the instantiation per LAUNCH by the
// filter's layout and the scan's length (a shorter scan runs the general one); the LAYOUT = 0 instantiations keep, textually, the
// code they had.
template <int LAYOUT> struct FixedLayout { static constexpr int S = 0, C = 0, MM = 0, CN = 0; };
template <> struct FixedLayout<1> { static constexpr int S = 1024, C = 512, MM = 64, CN = 256; };
template <> struct FixedLayout<2> { static constexpr int S = 512, C = 128, MM = 32, CN = 256; };
// LAYOUT 3 / 4 (round 6): the layouts of 1 / 2 WITHOUT the scan length — the measurement count comes from the arguments.  What a filter
// of one of these layouts runs on a scan of any other length than its capacity, i.e. on every real scan (the bundled data: 14 ... 44
// measurements): same visit, 4096 x 256 with scans of 27 / 44 / 61 measurements +3.3 / +3.5 / +2.9 % against the general instantiation,
// 16 384 particles +5.4 %, 256 x 64 with 13 / 27 measurements +4.3 / +2.8 %, and the CPHD variant **+24.8 / +14.6 / +14.3 %** (its
// general instantiation carries the cardinality rows' length and every table offset in registers) — profiles/r06_ab_layout_dynm.txt.
// A full scan keeps the instantiations that have its length too (1 / 2): -3.6 % (PHD), -8.8 % (CPHD) without it.
template <> struct FixedLayout<3> : FixedLayout<1> {};
template <> struct FixedLayout<4> : FixedLayout<2> {};
#define PHD_LAYOUT_FULL_SCAN(L) ((L) == 1 || (L) == 2)
#define PHD_A_SCAP (LAYOUT ? FixedLayout<LAYOUT>::S : A.S_cap)
#define PHD_A_CAP (LAYOUT ? FixedLayout<LAYOUT>::C : A.cap)
#define PHD_A_MM (LAYOUT ? FixedLayout<LAYOUT>::MM : A.MM)
// (CPHD: the length of the cardinality rows — max_cardinality + 1 = 256 in BASELINE.json configs[4] — a constant too: 3 040 -> 3 127
//  steps/s on top of the full scan; before the scan's length was a constant the same change measured -0.7 %)
#define PHD_A_CNLEN (LAYOUT ? FixedLayout<LAYOUT>::CN : A.cn_len)
template <bool STAMPS, bool FUSEW, bool CPHD, bool SPILL, int MINW = PHD_MIN_WAVES, bool GRIDT = false, int LAYOUT = 0>
__global__ __launch_bounds__(PHD_T, MINW) void phd_update_merge_kernel(UpdateArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Lds L = lds_carve((lds_u8)lds_raw, PHD_A_SCAP, PHD_A_CAP, PHD_A_MM);
    // CPHD instantiation: its arrays follow the common layout
    const CphdLds Q = CPHD ? cphd_carve((lds_u8)lds_raw + lds_offsets(PHD_A_SCAP, PHD_A_CAP, PHD_A_MM).total, (lds_u8)lds_raw, PHD_A_SCAP, PHD_A_CNLEN, PHD_A_MM) : CphdLds();
    if (CPHD) L.zpart = Q.zscr;   // the block parks rows of cn_len doubles where pass 1 leaves its partial sums: a scratch of its own

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
#ifdef PHD_EXP_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#define PHD_TRACE_END() do { if (A.trace && tid == 0) { A.trace[8 * blockIdx.x] = tr_t0; A.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); \
        A.trace[8 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } } while (0)
#define PHD_TRACE_AT(k) do { if (A.trace && tid == 0) A.trace[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PHD_TRACE_HANDOFF() do { if (A.trace) A.trace[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PHD_TRACE_END() do {} while (0)
#define PHD_TRACE_HANDOFF() do {} while (0)
#define PHD_TRACE_AT(k) do {} while (0)
#endif
    constexpr bool GRID_TAIL = GRIDT;
    if (FUSEW && (GRID_TAIL ? (int)blockIdx.x >= A.wa.n : blockIdx.x == gridDim.x - 1)) {
        // ---- fused step: the weights / nEff / resample routine has a workgroup of its own (the grid is N + 1).
        // It needs the particles' log-weight increments, predicted poses and map indirection — all known once a
        // particle's workgroup is past pass 1 — but not the merged maps (resampling moves indices, not maps).  So it
        // runs WHILE the slower half of every particle's work, the merge, is still going on, and its 4 us vanish
        // from the step.  Hand-off (cdna guide, Guideline 16): lane 0 of a particle's workgroup makes its hand-off
        // stores agent-scope, drains them and adds one to the ticket counter; this workgroup polls the counter
        // (agent scope, bounded) and then reads with sc1 loads.  Only this workgroup ever waits, and for workgroups
        // that wait for nothing, so the order in which the dispatcher places them does not matter.
        // Above 4096 particles the routine is the block form on several workgroups (phd_weights.h: weights_grid_body) - the grid is
        // N + W; each of them waits for the particles' tickets itself, and the LAST to leave re-arms the counters.  The dispatcher
        // starts workgroups in index order, so these take slots that the end of the launch would leave empty anyway; and should
        // they start early they hold slots, not progress: a particle's workgroup never waits for them.
        const unsigned n_wg = GRID_TAIL ? (unsigned)A.wa.n : gridDim.x - 1;
        const bool grid_form = GRID_TAIL;
        __shared__ int s_tail_ok;
        if (tid == 0) {
            unsigned spins = 0;
            bool ok = true;
            while (__hip_atomic_load(A.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_wg) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 24)) { ok = false; break; }   // ~seconds: never in practice
            }
#ifdef PHD_HANDOFF_MM
            // acquire: pairs with the release of every particle's ticket increment — what those workgroups stored before
            // it (pose, indirection, log-weight increment) is visible to the loads below
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            // time-out: the particles' workgroups are still adding to the ticket, so it is NOT reset here and the routine
            // does NOT run on incomplete data; the status bit makes the host fail the step and re-zero the ticket
            // (phd_device_status / phd_step_report)
            // (the block form re-arms the ticket itself: by the last of its workgroups to leave)
            if (ok) { if (!grid_form) __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // ready for the next launch
            else {
                atomicOr(A.status, PHD_STATUS_TAIL_TIMEOUT);
                if constexpr (GRID_TAIL)
                    if (A.wa.gsync) atomicCAS(&A.wa.gsync[3], 0u, 1u | (__hip_atomic_load(A.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8));
            }
            s_tail_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_tail_ok) return;
        // everything the routine needs is there: from here on it is the step's critical path whenever the last particles hand over
        // late (it shares its CU with up to three particle workgroups, which have the rest of their merge to hide behind it)
        __builtin_amdgcn_s_setprio(3);
        // the instantiation launch_weights runs for this particle count (one table — 256 threads up to 512 particles, 512 up to 4096: the
        // reduction trees, hence the bits of the log-sum-exp and of nEff, are a function of the block size), so
        // phd_step_dev and the staged calls agree bit for bit.  Up to 512 particles that is four waves; the other
        // four leave first: the routine's barriers then wait for half as many waves
        const bool shrink = A.wa.n_new <= A.wa.n;   // (the register-resident routine commits logw[j], j < n)
        if (grid_form) {
            // (the GRIDT instantiation only: see the template)
            if constexpr (GRID_TAIL) weights_grid_body<true>(A.wa, (int)blockIdx.x - A.wa.n, (int)gridDim.x - A.wa.n, lds_raw, A.status, A.ticket);
        } else if (A.wa.n <= 256 && shrink) {
            if (tid < 256) weights_body<256, 1, true>(A.wa, lds_raw);   // one weight per thread
        } else if (A.wa.n <= 512 && shrink) {
            if (tid < 256) weights_body<256, 2, true>(A.wa, lds_raw);
        } else if (A.wa.n <= 2 * PHD_T) {
            weights_body<PHD_T, 2, true>(A.wa, lds_raw);
        } else {
            weights_body<PHD_T, 8, true>(A.wa, lds_raw);                 // up to 4096
        }
        PHD_TRACE_END();
        return;
    }
    const int p = blockIdx.x;
    const DevConfig& cfg = A.cfg;
    // (LAYOUT != 0: a FULL scan — as many measurements as the filter holds, what BASELINE.json's configurations are — so that the
    //  measurement count is a constant too: the grids of pass 1 and pass 2, the CPHD block's tile count and sweep lengths fold)
    const int cap = PHD_A_CAP, S_cap = PHD_A_SCAP, M = PHD_LAYOUT_FULL_SCAN(LAYOUT) ? PHD_A_MM : A.M;
    const int src = A.parent[p];
    const float* __restrict__ in = A.map_in + (size_t)src * 6 * cap;
    const unsigned rows_stride = FUSEW ? 0u : A.out_stride; // rows mode belongs to the multi-GPU step (never the fused tail)
    float* __restrict__ out = A.map_out + (size_t)p * (rows_stride ? rows_stride : (size_t)6 * cap);
    const int n_map = A.count_in[src];
    // survivors past the LDS capacity go to this particle's record list in HBM (when the filter was created with one)
    const SpillRef sp = {SPILL ? A.spill_rec + (size_t)p * 2 * A.spill_cap * 8 : nullptr, SPILL ? A.spill_cap : 0};
    if (SPILL && tid == 0) A.spill_meta[(size_t)p * 8] = 0;
    phd_pose pose = A.pose[p];
    if (A.do_predict && wave == 0) {
        // fused vehicle predict: ONE wave computes the pose (a tangent, a sine / cosine pair, the generator: ~200 vector
        // instructions — seven more copies of them were 5 % of the workgroup's instructions when every lane computed its own)
        // and leaves it in LDS for the others, who pick it up behind the first barrier; lane 0 stores it to HBM there — a live
        // filter predicts in place, and every lane must have read the prior pose by then
        float n_alpha, n_encoder;
        if (A.noise) { n_alpha = A.noise[p].n_alpha; n_encoder = A.noise[p].n_encoder; }
        else draw_noise(A.seed, A.counter, p, cfg, n_alpha, n_encoder);
        pose = predict_pose(pose, A.control, n_alpha, n_encoder, cfg);
        if (lane == 0) { L.red[0] = pose.px; L.red[1] = pose.py; L.red[2] = pose.ptheta; }
    }
    u64* st = STAMPS ? (A.stamps + (size_t)p * PHD_STAMP_ROW) : nullptr;
    u64 cq[5] = {0, 0, 0, 0, 0};
    if (STAMPS && tid == 0) { for (int k = 12; k < PHD_STAMP_ROW; ++k) st[k] = 0; }
    STAMP(0);

    if (tid < 32) L.ctr[tid] = 0;
    const BucketMap bm = bucket_map(cfg.minFeatureWeight, S_cap);   // the merge's counting sort counts at emission (phd_lds.h)
    if (!CPHD) for (int b = tid; b < S_cap; b += PHD_T) ((LDS_T(u32)*)L.tr)[b] = 0u;   // (CPHD: the block below parks its rows here and clears afterwards)
    // measurements -> LDS (the reference keeps them in __constant__ Z[256], src/phdfilter.cu:120)
    for (int m = tid; m < M; m += PHD_T) {
        phd_measurement z = A.z[m];
        L.z_r[m] = z.range;
        L.z_b[m] = z.bearing;
        L.zok[m] = (z.label == 0 || !cfg.labeledMeasurements) ? 1u : 0u; // :1913
    }
    __syncthreads();
    // every thread holds the prior pose and src by now
    if (A.do_predict) { pose.px = L.red[0]; pose.py = L.red[1]; pose.ptheta = L.red[2]; pose.vx = 0.f; pose.vy = 0.f; pose.vtheta = 0.f; }
    if (tid == 0) {
        if (A.do_predict) {
            if (FUSEW) { // handed to the weights workgroup: agent-scope (sc1) stores
                float* po = (float*)&A.pose_out[p];
                __hip_atomic_store(po + 0, pose.px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(po + 1, pose.py, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(po + 2, pose.ptheta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(po + 3, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(po + 4, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(po + 5, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                A.pose_out[p] = pose;
            }
        }
        // (fused step) the output slab of particle p is its own again — a hand-off store
        if (FUSEW && A.parent_reset) __hip_atomic_store(&A.parent_reset[p], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- birth geometry (host loop src/phdfilter.cu:3470-3506): depends only on the pose and the measurement; thread m
    //      computes measurement m's now (M <= 256 < PHD_T: one each) and keeps the five numbers in registers until the
    //      normalisers are known — the lanes are idle here anyway, and LDS has no room for them (phd_lds.h)
    float bg_mx = 0.f, bg_my = 0.f, bg_xx = 0.f, bg_xy = 0.f, bg_yy = 0.f;
    static_assert(PHD_MAX_MEASUREMENTS <= PHD_T, "one birth per thread");
    // (PHD: the LAST waves take the births — thread 448 + m at up to 64 measurements: wave 0 has the predict and the hand-off, and at a
    //  few hundred particles a workgroup's critical path is the step (61.2 k -> 62.2 k steps/s at 256 x 64 x 32); same lanes as if wave
    //  0 took them, so the reduction of sum_m log Z_m adds the same numbers in the same order.  CPHD: the first waves — the last one
    //  finishes the CPHD block last (2 596 against 2 537 steps/s))
    const int birth_m = CPHD ? tid : tid - (PHD_T - 64 * ((M + 63) >> 6));
    const bool birth_thread = birth_m >= 0 && birth_m < M;
#ifndef PHD_EXP_BG_LATE
    if (CPHD) { if (tid < M) birth_geometry(pose, L.z_r[tid], L.z_b[tid], cfg, bg_mx, bg_my, bg_xx, bg_xy, bg_yy); }
    else if (birth_thread) birth_geometry(pose, L.z_r[birth_m], L.z_b[birth_m], cfg, bg_mx, bg_my, bg_xx, bg_xy, bg_yy);
#endif

    // ---- classification + per-feature EKF terms -----------------------------------------------
    float pdw_local = 0.f; // sum_j pd_j w_j (cardinality_predict, :2160)
    float wall_local = 0.f; // CPHD: <1, map>
    int n_in = 0, n_out0 = 0; // every thread accumulates the same totals: no broadcast (and no extra barrier) afterwards
    {
        // (the slab is requested for every slot below its CAPACITY, not below the map's count: the count is itself a load that
        //  depends on the indirection — parent[p] -> count[src] -> slab — and a lone workgroup per CU pays every link of that
        //  chain; the slab's address needs only `src`, so its loads now travel beside the count's.  At most cap - n_map unused
        //  words per plane, inside the slab)
        //  -DPHD_SLAB_AHEAD; built into no part at present (csrc/Makefile): same-visit A/B at 4096 x 256 x 64 while the kernels took their
        //  layout from the arguments: CPHD 2 640 -> 2 707 steps/s, the 80-register PHD kernel 4 263 -> 4 209 (its register allocation
        //  moves), 256 x 64 x 32 +0.3 %; with the layout and the scan's length compiled in: CPHD -1.3 %, the headline -0.5 %.
#ifdef PHD_SLAB_AHEAD
        int i0 = 0;
        do {
#else
        for (int i0 = 0; i0 < n_map; i0 += PHD_T) {
#endif
            const int i = i0 + tid;
            int cls = -1, nd_j = 0;
            float w = 0, mx = 0, my = 0, pxx = 0, pxy = 0, pyy = 0;
            EkfTerms t;
            t.pd = 0.f;
#ifdef PHD_SLAB_AHEAD
            if (i < cap) {
                w = in[0 * cap + i]; mx = in[1 * cap + i]; my = in[2 * cap + i];
                pxx = in[3 * cap + i]; pxy = in[4 * cap + i]; pyy = in[5 * cap + i];
            }
            if (i < n_map) {
#else
            if (i < n_map) {
                w = in[0 * cap + i]; mx = in[1 * cap + i]; my = in[2 * cap + i];
                pxx = in[3 * cap + i]; pxy = in[4 * cap + i]; pyy = in[5 * cap + i];
#endif
                ekf_terms(mx, my, pxx, pxy, pyy, pose, cfg, t);
                wall_local += w;
                // computeInRangeKernel, src/phdfilter.cu:1333-1346 (0.8/1.2 are double literals)
                const float ab = fabsf(t.b);
                if (t.r >= cfg.minRange && t.r <= cfg.maxRange && ab <= cfg.maxBearing) cls = 1;
                else if ((double)t.r >= 0.8 * (double)cfg.minRange && (double)t.r <= 1.2 * (double)cfg.maxRange &&
                         (double)ab <= 1.2 * (double)cfg.maxBearing) cls = 2;
                else cls = 0;
            }
            const u64 b_in = __ballot(cls == 1), b_out = __ballot(cls == 0);
            if (lane == 0) { L.ctr[CTR_TMP + wave] = __popcll(b_in); L.ctr[CTR_TMP + PHD_NW + wave] = __popcll(b_out); }
            __syncthreads();
            int off_in = n_in, off_out = n_out0, tot_in = 0, tot_out = 0;
            for (int wv = 0; wv < PHD_NW; ++wv) {
                const int ci = L.ctr[CTR_TMP + wv], co = L.ctr[CTR_TMP + PHD_NW + wv];
                if (wv < wave) { off_in += ci; off_out += co; }
                tot_in += ci; tot_out += co;
            }
            if (cls == 1) {
                const int j = off_in + __popcll(b_in & lanemask_lt());
                // log pd + log w + (-log 2pi - 0.5 log det)   (:1911,1916-1917), folded per feature
                const float lw0 = safe_log(t.pd) + safe_log(w);
                L.f_a[j] = (v4f){t.r, t.b, t.s00, t.s12};
                L.f_c[j] = (v2f){t.s11, lw0 - safe_log(6.2831855f) - 0.5f * safe_log(t.det)};
                L.f_idx[j] = (u16)i;
                // (gain and Joseph covariance, src/phdfilter.cu:1884-1901, are rebuilt for the detection terms that survive
                //  the prune — detection_posterior() — instead of being kept for every feature: 36 B of LDS per feature)
                pdw_local += t.pd * w;
                nd_j = j;
            } else if (cls == 0) {
                L.out_idx[off_out + __popcll(b_out & lanemask_lt())] = (u16)i;
            }
            // straight into the survivor list: nearly-in-range features, which skip the update and join
            // the merge (:3242-3257), and the non-detection term of an in-range feature — the prior
            // with weight w(1-pd) (:2145-2148) — unless it is pruned (:2314)
            if (CPHD) {
                // the CPHD weights carry the factor <Y1,p>/<Y0,p>, known after the ESFs: emission is deferred;
                // remember the nearly-in-range features at the top of out_idx
                if (cls == 2) L.out_idx[cap - 1 - atomicAdd((int*)&L.ctr[CTR_NNEAR], 1)] = (u16)i;
            } else {
                const float wnd = w * (1 - t.pd);
                const bool keep = (cls == 2) || (cls == 1 && !(wnd < cfg.minFeatureWeight));
                const int slot = alloc_slots(keep, L.ctr);
                if (keep) store_survivor(L, slot, S_cap, cls == 2 ? w : wnd, mx, my, pxx, pxy, pyy,
                                         cls == 2 ? NEAR_U_BASE + i : nd_j, sp, bm);
            }
            n_in += tot_in; n_out0 += tot_out;
            __syncthreads();
#ifdef PHD_SLAB_AHEAD
        } while ((i0 += PHD_T) < n_map);
#else
        }
#endif
    }
    STAMP(1);

    // lane <-> measurement mapping
    int Mp = 1;
    while (Mp < M && Mp < 64) Mp <<= 1;          // lanes per feature group
    const int JS = 64 / Mp;                      // features in flight per wave
    const int m_tiles = (M + 63) / 64;
    const int lm = lane & (Mp - 1);
    const int js = lane / Mp;

    // Pass 2 keeps a detection term iff exp(lw - log Z_m) >= minFeatureWeight, and log Z_m >= log(clutter + birth
    // weight) whatever the map: a term with lw below log(minFeatureWeight) + log(clutter + birth) (1e-3 of slack for the
    // rounding of either side) cannot survive.  Pass 1 has lw in a register, so it lists the few terms that can
    // (about 1 in 10) and pass 2 visits the list, one term per thread, instead of every (feature, measurement)
    // pair again.  A full list falls back to the dense pass.
    // CPHD: the weight is e_jm (lambda/kappa) <Y1[Z\m],p> / <Y0[Z],p>.  With e_j(Z) = e_j(Z\m) + xi_m e_{j-1}(Z\m) >=
    // xi_m e_{j-1}(Z\m) and the coefficient identity c1_j = c0_{j+1} (I_1[j] = I_0[j+1], cphd_block), <Y0[Z],p> >=
    // xi_m <Y1[Z\m],p>, xi_m = (lambda/kappa)(S_m + birth weight): the factor is at most 1/(S_m + birth) <= 1/birth,
    // so lw >= log(minFeatureWeight) + log(birth) is necessary there (5e-2 of slack for the rounding of the ESF chain).
    lds_u16 clist = (lds_u16)L.part;
    const bool sparse2 = cfg.minFeatureWeight > 0.f && (n_in * M <= 0xFFFF);
    const float c0m = CPHD ? safe_log(cfg.minFeatureWeight) + safe_log(cfg.birthWeight) - 5e-2f
                           : safe_log(cfg.minFeatureWeight) + safe_log(cfg.clutterDensity + cfg.birthWeight) - 1e-3f;
    // ---- pass 1: normalisers (phd_pass1.h) -----------------------------------------------------------
    const Pass1Grid p1g = pass1_grid(M);
    pass1_normalisers(L, n_in, M, PHD_A_MM, tid, sparse2, c0m);
#ifdef PHD_DUP_PASS1   // (tools/ab_bench.sh: the sweep run twice - its marginal cost; the candidate list restarted, the sums rewritten)
    __syncthreads();
    if (tid == 0) L.ctr[CTR_NCAND] = 0;
    __syncthreads();
    pass1_normalisers(L, n_in, M, PHD_A_MM, tid, sparse2, c0m);
#endif
    if (STAMPS && tid == 0) st[23] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (STAMPS && tid == 0) st[24] = __builtin_amdgcn_s_memrealtime();
    float lz_local = 0.f;
    float dl_phd = 0.f;     // the particle's log-weight increment (PHD), every thread the same; handed over after pass 2
    if (CPHD) {
        // roots Xi_m = (lambda/kappa)(sum_j pd w_j g_jm + birthWeight) (.bak:1205-1222)
        const float rat = cfg.clutterRate / cfg.clutterDensity;
        for (int m = tid; m < M; m += PHD_T) Q.lxi[m] = (pass1_feature_sum(L, p1g, m, M) + cfg.birthWeight) * rat;
        const float pdw = block_sum(pdw_local, L.red, tid);
        const float w_all = block_sum(wall_local, L.red, tid);
        __syncthreads();
        cphd_block(L, Q, cfg, M, PHD_A_MM, PHD_A_CNLEN, A.lfact, A.lfact_len, A.cn_in + (size_t)src * PHD_A_CNLEN,
                   A.cn_out + (size_t)p * (rows_stride ? rows_stride : (unsigned)PHD_A_CNLEN), A.cphd_scratch + (size_t)p * PHD_A_MM * PHD_A_MM, w_all, pdw, tid,
                   STAMPS ? cq : nullptr, S_cap, STAMPS ? st : nullptr);
        const float r1 = Q.scal[CQ_R1];
        // births (weight bw (lambda/kappa) <Y1[Z\m],p>/<Y0,p>)
        for (int m0 = 0; m0 < M; m0 += PHD_T) {
            const int m = m0 + tid;
            const bool mv = m < M;
            const float wb = (mv && L.zok[mv ? m : 0]) ? expf(safe_log(cfg.birthWeight) - L.logZ[mv ? m : 0]) : 0.f;
            const bool keep = mv && !(wb < cfg.minFeatureWeight);
            const int slot = alloc_slots(keep, L.ctr);
#ifdef PHD_EXP_BG_LATE
            if (keep) birth_geometry(pose, L.z_r[m], L.z_b[m], cfg, bg_mx, bg_my, bg_xx, bg_xy, bg_yy);
#endif
            if (keep) store_survivor(L, slot, S_cap, wb, bg_mx, bg_my, bg_xx, bg_xy, bg_yy, n_in + M * n_in + m, sp, bm);
        }
        // missed detections of the in-range features: w (1 - pd) r1 (.bak:1445-1460)
        for (int j0 = 0; j0 < n_in; j0 += PHD_T) {
            const int j = j0 + tid;
            const bool jv = j < n_in;
            const int i = L.f_idx[jv ? j : 0];
            const float wnd = jv ? in[0 * cap + i] * (1 - cfg.pd) * r1 : 0.f;
            const bool keep = jv && !(wnd < cfg.minFeatureWeight);
            const int slot = alloc_slots(keep, L.ctr);
            if (keep) store_survivor(L, slot, S_cap, wnd, in[1 * cap + i], in[2 * cap + i], in[3 * cap + i], in[4 * cap + i],
                                     in[5 * cap + i], j, sp, bm);
        }
        // nearly-in-range features (pD = 0): weight w r1, join the merge unpruned like HEAD (:3242-3257)
        const int n_near = L.ctr[CTR_NNEAR];
        for (int k0 = 0; k0 < n_near; k0 += PHD_T) {
            const int k = k0 + tid;
            const bool kv = k < n_near;
            const int i = L.out_idx[cap - 1 - (kv ? k : 0)];
            const int slot = alloc_slots(kv, L.ctr);
            if (kv) store_survivor(L, slot, S_cap, in[0 * cap + i] * r1, in[1 * cap + i], in[2 * cap + i], in[3 * cap + i],
                                   in[4 * cap + i], in[5 * cap + i], NEAR_U_BASE + i, sp, bm);
        }
        // (the particle's log-weight increment, log <Y0,p>, .bak:2661-2667, is handed over after pass 2 — see there)
    } else {
    if (birth_thread) {
        const int m = birth_m;
        float sum = pass1_feature_sum(L, p1g, m, M);
        sum += cfg.clutterDensity;                                                                    // :2213
        sum += cfg.birthWeight;                                                                       // :2214
        const float lz = safe_log(sum);                                                               // :2217
        L.logZ[m] = lz;
        lz_local += lz;                                                                               // :2251
        // the birth term of this measurement (weight :2239-2243, prune :2314) while lz is in a register
        const float wb = L.zok[m] ? expf(safe_log(cfg.birthWeight) - lz) : 0.f;
        const bool keep = !(wb < cfg.minFeatureWeight);
        const int slot = alloc_slots(keep, L.ctr);
#ifdef PHD_EXP_BG_LATE
        if (keep) birth_geometry(pose, L.z_r[m], L.z_b[m], cfg, bg_mx, bg_my, bg_xx, bg_xy, bg_yy);
#endif
        if (keep) store_survivor(L, slot, S_cap, wb, bg_mx, bg_my, bg_xx, bg_xy, bg_yy, n_in + M * n_in + m, sp, bm);
    }
    {
        float lz_sum, pdw;
        block_sum2(lz_local, pdw_local, L.red, tid, lz_sum, pdw, /*scratch_idle=*/true); // the only reduction of the PHD path
        // particle_weighting == 0 (:2260-2263): sum_m log Z_m - (sum_j pd_j w_j + M * birthWeight)
        dl_phd = lz_sum - (pdw + (float)M * cfg.birthWeight);
    }
    } // !CPHD
    // PHD: everything pass 2 reads (log Z_m, the candidate list and its counts) was published by the barriers above
    if (CPHD) __syncthreads();

    STAMP(2);
    // ---- pass 2: final weights; prune before store --------------------------------------------------
    STAMP(3);
    // detection terms
    const int n_cand = sparse2 ? L.ctr[CTR_NCAND] : 0;
    if (sparse2 && n_cand <= PHD_CAND_CAP) {
        const float rcp_nin = 1.0f / (float)(n_in > 0 ? n_in : 1);
        for (int t0 = 0; t0 < n_cand; t0 += PHD_T) {
            const int t = t0 + tid;
            const bool tv = t < n_cand;
            const int idx = tv ? clist[t] : 0;
            const int m = (int)(((float)idx + 0.5f) * rcp_nin);      // idx = m * n_in + j < 2^16: exact
            const int j = idx - m * n_in;
            const v4f fa = L.f_a[j];
            const v2f fc = L.f_c[j];
            const float i0 = L.z_r[m] - fa.x;
            const float i1 = wrap_angle(L.z_b[m] - fa.y);
            const float dist = i0 * i0 * fa.z + i0 * i1 * fa.w + i1 * i1 * fc.x;
            const float lw = fc.y - 0.5f * dist;
            const float w = __expf(lw - L.logZ[m]);                                                   // :2242-2243 (listed terms have a valid label)
            const bool keep = tv && !(w < cfg.minFeatureWeight);                                      // :2314
            const int slot = alloc_slots(keep, L.ctr);
            if (keep) {
                // the complete survivor at once (two thirds of the listed terms survive: a separate finalise pass would
                // fetch the same feature terms again, behind one more barrier)
                float umx, umy, uxx, uxy, uyy;
                detection_posterior(in, cap, L.f_idx[j], pose, cfg, fa.x, fa.z, fa.w, fc.x, i0, i1, umx, umy, uxx, uxy, uyy);
                if (slot < S_cap) {
                    L.w[slot] = w; L.u[slot] = n_in + m * n_in + j;
                    L.mx[slot] = umx; L.my[slot] = umy;
                    L.xx[slot] = uxx; L.xy[slot] = uxy; L.yy[slot] = uyy;
                    __hip_atomic_fetch_add((LDS_T(u32)*)L.tr + bucket_of(bm, w), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                else {   // past the LDS capacity: the complete record goes to the spill list (no finalise pass there)
                    spill_store(sp, L, slot, w, umx, umy, uxx, uxy, uyy, n_in + m * n_in + j);
                }
            }
        }
    } else
    for (int mt = 0; mt < m_tiles; ++mt) {
        const int m = mt * 64 + lm;
        const bool mvalid = (m < M);
        const bool zok = mvalid && L.zok[mvalid ? m : 0];
        const float zr = L.z_r[mvalid ? m : 0], zb = L.z_b[mvalid ? m : 0], lz = L.logZ[mvalid ? m : 0];
        for (int jb = wave * JS; jb < n_in; jb += PHD_NW * JS) {
            const int j = jb + js;
            const int jj = j < n_in ? j : n_in - 1;
            const v4f fa = L.f_a[jj];
            const v2f fc = L.f_c[jj];
            const float i0 = zr - fa.x;
            const float i1 = wrap_angle(zb - fa.y);
            const float dist = i0 * i0 * fa.z + i0 * i1 * fa.w + i1 * i1 * fc.x;
            const float lw = fc.y - 0.5f * dist;
            const float w = zok ? __expf(lw - lz) : 0.f;                                              // :2242-2243
            const bool keep = mvalid && (j < n_in) && !(w < cfg.minFeatureWeight);                    // :2314
            // survivors are sparse (~1 %): the hot loop only records (weight, slab index); mean and
            // covariance are filled in by the dense finalise pass below, one lane per survivor
            const int slot = alloc_slots(keep, L.ctr);
            if (keep) {
                if (slot < S_cap) {
                    L.w[slot] = w; L.u[slot] = n_in + m * n_in + j;
                    __hip_atomic_fetch_add((LDS_T(u32)*)L.tr + bucket_of(bm, w), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                else {
                    float umx, umy, uxx, uxy, uyy;
                    detection_posterior(in, cap, L.f_idx[jj], pose, cfg, fa.x, fa.z, fa.w, fc.x, i0, i1, umx, umy, uxx, uxy, uyy);
                    spill_store(sp, L, slot, w, umx, umy, uxx, uxy, uyy, n_in + m * n_in + j);
                }
            }
        }
    }
    // ---- the particle's log-weight increment, and (fused step) the hand-off to the weights workgroup.  Known since pass 1, stored
    //      only now: vector-memory operations complete IN ORDER for a wave's counter, so a store or the ticket's atomic issued
    //      before pass 2 makes this wave's loads of the prior features (detection_posterior) wait for the store's round trip to
    //      memory — 1 to 2.5 us depending on the XCD, with seven waves waiting at the barrier (measured with workgroup time
    //      stamps, profiles/r04_three_workgroups.txt).  After pass 2 nothing of this wave follows the ticket until the tail.
    if (tid == 0) {
        const float dl = CPHD ? Q.scal[CQ_LY0] : dl_phd;
        if (FUSEW) {
            // the last of this particle's hand-off stores (pose: at the top; indirection: after the first barrier)
            __hip_atomic_store(&A.dlogw[p], dl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            handoff_ticket(A.ticket);
            PHD_TRACE_HANDOFF();
        } else {
            A.dlogw[p] = dl;
            if (A.raw_out) A.raw_out[p] = A.logw_in[p] + dl;                                          // :3741-3744
        }
    }
    STAMP(4);
    PHD_TRACE_AT(6);
    const bool finalised = sparse2 && n_cand <= PHD_CAND_CAP;     // (uniform) the listed pass stored complete survivors
    if (!finalised) __syncthreads();
    if (!finalised) {
        // dense finalise of the detection terms: rebuild gain, updated mean and Joseph covariance
        // from the prior feature (src/phdfilter.cu:1884-1906)
        const int n_now = min(L.ctr[CTR_NSURV], S_cap);
        const int u_det_lo = n_in, u_det_hi = n_in + M * n_in;
        for (int s = tid; s < n_now; s += PHD_T) {
            const int u = L.u[s];
            if (u < u_det_lo || u >= u_det_hi) continue;
            const int m = (u - n_in) / n_in;
            const int j = (u - n_in) - m * n_in;
            const v4f fa = L.f_a[j];
            const float i0 = L.z_r[m] - fa.x;
            const float i1 = wrap_angle(L.z_b[m] - fa.y);
            float umx, umy, uxx, uxy, uyy;
            detection_posterior(in, cap, L.f_idx[j], pose, cfg, fa.x, fa.z, fa.w, L.f_c[j].x, i0, i1, umx, umy, uxx, uxy, uyy);
            L.mx[s] = umx; L.my[s] = umy;
            L.xx[s] = uxx; L.xy[s] = uxy; L.yy[s] = uyy;
        }
    }
    __syncthreads();
    STAMP(5);
    PHD_TRACE_AT(4);
    int n_surv = L.ctr[CTR_NSURV];
    unsigned status = 0;
    // more survivors than LDS holds: with a spill list (and room in it) the particle's merge is handed to
    // phd_merge_spill_kernel; without one the list is truncated and the step reports PHD_ERR_CAPACITY
    const bool spilled = SPILL && n_surv > S_cap && n_surv <= sp.cap && !L.ctr[CTR_OVERFLOW];
    const int n_all = n_surv;
    if (n_surv > S_cap) { n_surv = S_cap; if (!spilled) status |= PHD_STATUS_SURVIVOR_OVERFLOW; }
    if (L.ctr[CTR_OVERFLOW]) status |= PHD_STATUS_SURVIVOR_OVERFLOW;
    if (spilled) {
        // the LDS-resident survivors join the records (list position = arrival slot), then the hand-over data
        for (int i = tid; i < S_cap; i += PHD_T) {
            float* r = sp.rec + (size_t)i * 8;
            r[0] = L.w[i]; r[1] = L.mx[i]; r[2] = L.my[i]; r[3] = L.xx[i]; r[4] = L.xy[i]; r[5] = L.yy[i];
            r[6] = __int_as_float(L.u[i]); r[7] = 0.f;
        }
        unsigned short* oi = A.spill_out + (size_t)p * cap;
        for (int i = tid; i < n_out0; i += PHD_T) oi[i] = L.out_idx[i];
        if (tid == 0) {
            int* meta = A.spill_meta + (size_t)p * 8;
            meta[1] = n_in * (M + 1) + M;
            meta[2] = n_out0;
            meta[3] = __float_as_int(CPHD ? Q.scal[CQ_R1] : 1.f);
            // the input slab: the indirection parent[p] is reset to p by this launch (parent_reset aliases parent), so the
            // spill kernel cannot look it up again after a resample left parent[p] != p
            meta[4] = src;
            meta[0] = n_all;
        }
    }

    // optional inspection copy of the survivors (parity tests)
    if (A.dbg_surv) {
        const int dc = A.dbg_cap;
        float* d = A.dbg_surv + (size_t)p * 6 * dc;
        int* du = A.dbg_u + (size_t)p * dc;
        const int n_dbg = spilled ? (n_all < dc ? n_all : dc) : n_surv;
        __syncthreads();   // (the records of the spill list written above, by other threads)
        for (int i = tid; i < n_dbg; i += PHD_T) {
            if (i < S_cap) {
                d[0 * dc + i] = L.w[i]; d[1 * dc + i] = L.mx[i]; d[2 * dc + i] = L.my[i];
                d[3 * dc + i] = L.xx[i]; d[4 * dc + i] = L.xy[i]; d[5 * dc + i] = L.yy[i];
                du[i] = L.u[i];
            } else {
                const float* r = sp.rec + (size_t)i * 8;
                d[0 * dc + i] = r[0]; d[1 * dc + i] = r[1]; d[2 * dc + i] = r[2];
                d[3 * dc + i] = r[3]; d[4 * dc + i] = r[4]; d[5 * dc + i] = r[5];
                du[i] = __float_as_int(r[6]);
            }
        }
        if (tid == 0) { A.dbg_n[p] = n_dbg; A.dbg_nin[p] = n_in; }
        __syncthreads(); // the merge permutes the planes the copy reads
    }
    if (spilled) {
        // everything but the merged map: indirection reset, export-row header, status, survivor high-water mark
        if (tid == 0) {
            if (!FUSEW && A.parent_reset) A.parent_reset[p] = p;
            if (rows_stride) {
                float* h = out - 8;
                const float* ps = (const float*)(A.do_predict ? &A.pose_out[p] : &A.pose[p]);
#pragma unroll
                for (int k = 0; k < 6; ++k) h[k] = ps[k];
                h[7] = A.raw_out[p];
            }
            if (status) atomicOr(A.status, status);
            if (n_all > __hip_atomic_load(A.max_surv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(A.max_surv, n_all);
        }
        STAMP(11);
        return;
    }

    // ---- merge ----------------------------------------------------------------------------------------
    const int n_update = n_in * (M + 1) + M;
    const bool packed = (n_update + n_map <= 0xFFFF) && (S_cap <= 0x10000);
    // (the PHD instantiations with a compiled-in layout also assume the Mahalanobis merge metric — the reference's default, and a
    //  per-filter constant: the Hellinger copy of the merge is not in them; +0.7 % at the headline, nothing for CPHD, which keeps both)
    if ((LAYOUT && !CPHD) || cfg.distanceMetric == 0) merge_in_lds<false, STAMPS>(L, S_cap, n_surv, cfg, out, cap, tid, st, n_update, packed, bm);
    else merge_in_lds<true, STAMPS>(L, S_cap, n_surv, cfg, out, cap, tid, st, n_update, packed, bm);
    PHD_TRACE_AT(5);
    int k_out = L.ctr[CTR_KOUT];
    if (k_out > cap) { k_out = cap; status |= PHD_STATUS_MAP_OVERFLOW; }
    // append the untouched out-of-range features (src/phdfilter.cu:3311-3318)
    int n_app = n_out0;
    if (k_out + n_app > cap) { n_app = cap - k_out; status |= PHD_STATUS_MAP_OVERFLOW; }
    const float r_out = CPHD ? Q.scal[CQ_R1] : 1.f; // CPHD: undetected mass outside the field of view takes r1 too
    for (int i = tid; i < n_app; i += PHD_T) {
        const int s = L.out_idx[i];
        out[k_out + i] = CPHD ? in[s] * r_out : in[s];
#pragma unroll
        for (int pl = 1; pl < 6; ++pl) out[pl * cap + k_out + i] = in[pl * cap + s];
    }
    if (tid == 0) {
        if (!FUSEW && A.parent_reset) A.parent_reset[p] = p; // the output slab of particle p is its own again
        A.count_out[p] = k_out + n_app;
        if (rows_stride) { // the export row's header (phd_export_kernel's layout)
            // pose and raw weight: re-read what this thread stored earlier rather than keep them live through the merge
            float* h = out - 8;
            const float* ps = (const float*)(A.do_predict ? &A.pose_out[p] : &A.pose[p]);
#pragma unroll
            for (int k = 0; k < 6; ++k) h[k] = ps[k];
            ((int*)h)[6] = k_out + n_app;
            h[7] = A.raw_out[p];
        }
        if (status) atomicOr(A.status, status);
        // high-water marks (diagnostics): a plain look first — in steady state no workgroup raises them, and 2 x N
        // same-address atomics per launch are not free
        const int hs = L.ctr[CTR_NSURV], hm = k_out + n_out0;
        if (hs > __hip_atomic_load(A.max_surv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(A.max_surv, hs);
        if (hm > __hip_atomic_load(A.max_map, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(A.max_map, hm);
    }
    STAMP(11);
    PHD_TRACE_END();
    if (STAMPS && CPHD && tid == 0) { // the CPHD block's parts replace the merge-round statistics
        st[12] = cq[1] - cq[0]; st[13] = cq[2] - cq[1]; st[14] = cq[3] - cq[2]; st[15] = cq[4] - cq[3];
    }
}

// ------------------------------------------------------------------------------------------
#undef PHD_A_SCAP
#undef PHD_A_CAP
#undef PHD_A_MM
#undef PHD_A_CNLEN
// Four translation units from this one file (csrc/Makefile): the CPHD instantiations of the update kernel and the
// three-workgroups-per-CU ones (PHD, CPHD) are compiled on their own with -DPHD_CPHD_TU / -DPHD_W6_TU / -DPHD_CPHD_W6_TU — the
// kernel template above, one of these tables and nothing else — so that the parts build side by side and the CPHD part can take compile flags of its own
// (csrc/Makefile, KFLAGS_CPHD: measured, currently the same).
// ------------------------------------------------------------------------------------------
#if defined(PHD_CPHD_W6_TU)
// <STAMPS, FUSEW, CPHD, -, 6>: the same three for the CPHD filter
// [3], [4]: the staged and the fused step with the layout of BASELINE.json's 256-Gaussian configurations compiled in (LAYOUT = 1)
// [6], [7], [8]: the same three with LAYOUT = 3 (the layout without the scan length: scans shorter than the filter's capacity)
extern const void* const k_update_cphd_w6_fns[9] = {(const void*)phd_update_merge_kernel<false, false, true, false, 6>,
                                                    (const void*)phd_update_merge_kernel<true, false, true, false, 6>,
                                                    (const void*)phd_update_merge_kernel<false, true, true, false, 6>,
                                                    (const void*)phd_update_merge_kernel<false, false, true, false, 6, false, 1>,
                                                    (const void*)phd_update_merge_kernel<false, true, true, false, 6, false, 1>,
                                                    (const void*)phd_update_merge_kernel<true, false, true, false, 6, false, 1>,    // [5]: the diagnostic one
                                                    (const void*)phd_update_merge_kernel<false, false, true, false, 6, false, 3>,
                                                    (const void*)phd_update_merge_kernel<false, true, true, false, 6, false, 3>,
                                                    (const void*)phd_update_merge_kernel<true, false, true, false, 6, false, 3>};
#elif defined(PHD_W6_TU)
// <STAMPS, FUSEW, -, -, 6>: the staged step, the diagnostic instantiation and the fused step for three workgroups per CU
// + the fused step with the block-form tail (more than 4096 particles)
// [4], [5], [6]: the staged step, the fused step (the headline) and the fused step with the block-form tail with LAYOUT = 1 compiled in
// [8] ... [11]: the same four with LAYOUT = 3 (the layout without the scan length)
extern const void* const k_update_w6_fns[12] = {(const void*)phd_update_merge_kernel<false, false, false, false, 6>,
                                               (const void*)phd_update_merge_kernel<true, false, false, false, 6>,
                                               (const void*)phd_update_merge_kernel<false, true, false, false, 6>,
                                               (const void*)phd_update_merge_kernel<false, true, false, false, 6, true>,
                                               (const void*)phd_update_merge_kernel<false, false, false, false, 6, false, 1>,
                                               (const void*)phd_update_merge_kernel<false, true, false, false, 6, false, 1>,
                                               (const void*)phd_update_merge_kernel<false, true, false, false, 6, true, 1>,
                                               (const void*)phd_update_merge_kernel<true, false, false, false, 6, false, 1>,    // [7]: the diagnostic one
                                               (const void*)phd_update_merge_kernel<false, false, false, false, 6, false, 3>,
                                               (const void*)phd_update_merge_kernel<false, true, false, false, 6, false, 3>,
                                               (const void*)phd_update_merge_kernel<false, true, false, false, 6, true, 3>,
                                               (const void*)phd_update_merge_kernel<true, false, false, false, 6, false, 3>};
#elif defined(PHD_CPHD_TU)
// <STAMPS, FUSEW, CPHD, SPILL>: the staged step, the diagnostic instantiation, the fused step, and the two with a spill list
extern const void* const k_update_cphd_fns[5] = {(const void*)phd_update_merge_kernel<false, false, true, false>,
                                                 (const void*)phd_update_merge_kernel<true, false, true, false>,
                                                 (const void*)phd_update_merge_kernel<false, true, true, false>,
                                                 (const void*)phd_update_merge_kernel<false, false, true, true>,
                                                 (const void*)phd_update_merge_kernel<false, true, true, true>};
#else
extern const void* const k_update_cphd_fns[5];
extern const void* const k_update_w6_fns[12];
extern const void* const k_update_cphd_w6_fns[9];

__global__ void phd_predict_kernel(const phd_pose* __restrict__ in, phd_pose* __restrict__ out, int n,
                                   phd_ackerman_control u, const phd_ackerman_noise* __restrict__ noise,
                                   u64 seed, u64 counter, DevConfig cfg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float n_alpha, n_encoder;
    if (noise) { n_alpha = noise[i].n_alpha; n_encoder = noise[i].n_encoder; }
    else draw_noise(seed, counter, i, cfg, n_alpha, n_encoder);
    out[i] = predict_pose(in[i], u, n_alpha, n_encoder, cfg);
}

// particle "shotgun" (n_predict_particles = k > 1; src/phdfilter.cu:797-823,1185-1238): predicted
// particle idx descends from prior particle idx / k, draws its own control noise, shares the prior's map
// through the parent indirection (the reference deep-copies the maps k times) and carries the prior's
// log-weight minus log k
__global__ void phd_predict_shotgun_kernel(const phd_pose* __restrict__ in, phd_pose* __restrict__ out, int n_pred, int k,
                                           phd_ackerman_control u, const phd_ackerman_noise* __restrict__ noise,
                                           u64 seed, u64 counter, DevConfig cfg, const int* __restrict__ parent_in,
                                           int* __restrict__ parent_out, const float* __restrict__ logw_in,
                                           float* __restrict__ logw_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pred) return;
    const int prior = i / k;                                                    // :797
    float n_alpha, n_encoder;
    if (noise) { n_alpha = noise[i].n_alpha; n_encoder = noise[i].n_encoder; }
    else draw_noise(seed, counter, i, cfg, n_alpha, n_encoder);
    out[i] = predict_pose(in[prior], u, n_alpha, n_encoder, cfg);
    parent_out[i] = parent_in[prior];
    logw_out[i] = logw_in[prior] - safe_log((float)k);                           // :1213
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
// AoS Gaussian2D (28 B, reference layout) <-> SoA slab planes [w, mx, my, pxx, pxy, pyy][cap]
__global__ void phd_pack_maps_kernel(const phd_gaussian2d* __restrict__ concat, const int* __restrict__ offsets,
                                     const int* __restrict__ sizes, float* __restrict__ slabs, int cap)
{
    const int p = blockIdx.x;
    const int n = sizes[p];
    const phd_gaussian2d* g = concat + offsets[p];
    float* s = slabs + (size_t)p * 6 * cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        phd_gaussian2d v = g[i];
        s[0 * cap + i] = v.weight;
        s[1 * cap + i] = v.mean[0];
        s[2 * cap + i] = v.mean[1];
        s[3 * cap + i] = v.cov[0];
        s[4 * cap + i] = (v.cov[1] + v.cov[2]) * 0.5f; // maps the filter produces are symmetric already
        s[5 * cap + i] = v.cov[3];
    }
}

__global__ void phd_unpack_maps_kernel(const float* __restrict__ slabs, const int* __restrict__ parent,
                                       const int* __restrict__ offsets, const int* __restrict__ counts,
                                       phd_gaussian2d* __restrict__ concat, int cap)
{
    const int p = blockIdx.x;
    const int src = parent ? parent[p] : p;
    const int n = counts[src];
    const float* s = slabs + (size_t)src * 6 * cap;
    phd_gaussian2d* g = concat + offsets[p];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        phd_gaussian2d v;
        v.weight = s[0 * cap + i];
        v.mean[0] = s[1 * cap + i];
        v.mean[1] = s[2 * cap + i];
        v.cov[0] = s[3 * cap + i];
        v.cov[1] = s[4 * cap + i];
        v.cov[2] = s[4 * cap + i];
        v.cov[3] = s[5 * cap + i];
        g[i] = v;
    }
}

// weighted-mean pose (src/main.cpp:331-340) and arg-max weight (:347-356); one workgroup
#define PHD_WT 1024
__global__ __launch_bounds__(PHD_WT) void phd_state_kernel(const phd_pose* __restrict__ poses,
                                                           const float* __restrict__ logw, int n,
                                                           float* __restrict__ pose_out, int* __restrict__ argmax_out)
{
    __shared__ float sc[PHD_WT / 64];
    __shared__ float s_best[PHD_WT / 64];
    __shared__ int s_besti[PHD_WT / 64];
    const int tid = threadIdx.x;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    float best = -FLT_MAX;
    int besti = 0x7FFFFFFF;
    for (int i = tid; i < n; i += PHD_WT) {
        const float lw = logw[i];
        const float w = expf(lw);
        const phd_pose q = poses[i];
        acc[0] += w * q.px; acc[1] += w * q.py; acc[2] += w * q.ptheta;
        acc[3] += w * q.vx; acc[4] += w * q.vy; acc[5] += w * q.vtheta;
        if (lw > best) { best = lw; besti = i; }
    }
    for (int k = 0; k < 6; ++k) {
        const float r = block_reduce_w<PHD_WT>(acc[k], sc, tid, false);
        if (tid == 0) pose_out[k] = r;
    }
    // arg-max with ties to the lowest index (strict '>' scan in the reference)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = xor_lane(best, off);
        const int oi = xor_lane(besti, off);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_besti[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < PHD_WT / 64; ++w)
            if (s_best[w] > best || (s_best[w] == best && s_besti[w] < besti)) { best = s_best[w]; besti = s_besti[w]; }
        if (n == 1) { // src/main.cpp:381-384
            pose_out[0] = poses[0].px; pose_out[1] = poses[0].py; pose_out[2] = poses[0].ptheta;
            pose_out[3] = poses[0].vx; pose_out[4] = poses[0].vy; pose_out[5] = poses[0].vtheta;
        }
        argmax_out[0] = (besti == 0x7FFFFFFF) ? -1 : besti;
    }
}

// gather/scatter of whole particles for peer migration: [pose (6 f32) | count (1 i32) | pad | 6*cap f32]
// [7] carries the particle's un-normalised log-weight when `raw` is given (whole-shard export of the gathered exchange)
__global__ void phd_export_kernel(const float* __restrict__ slabs, const int* __restrict__ counts,
                                  const int* __restrict__ parent, const phd_pose* __restrict__ poses,
                                  const int* __restrict__ which, unsigned char* __restrict__ buf, int cap, size_t stride,
                                  const float* __restrict__ raw)
{
    const int k = blockIdx.x;
    const int p = which ? which[k] : k;
    const int src = parent ? parent[p] : p;
    float* o = (float*)(buf + (size_t)k * stride);
    const int n = counts[src];
    if (threadIdx.x < 6) o[threadIdx.x] = ((const float*)&poses[p])[threadIdx.x];
    if (threadIdx.x == 6) ((int*)o)[6] = n;
    if (threadIdx.x == 7) o[7] = raw ? raw[p] : 0.f;
    const float* s = slabs + (size_t)src * 6 * cap;
    for (int i = threadIdx.x; i < 6 * cap; i += blockDim.x) o[8 + i] = ((i % cap) < n) ? s[i] : 0.f;
}

// which == NULL: slot k; rowsel != NULL: slot k takes row rowsel[k] of the buffer (the gathered exchange: every rank
// holds all rows, rowsel = this shard's part of the global parent indices, on the device); logw_fill/parent_reset: the
// tail of copy_particles (weights <- -log N, src/slamtypes.h:327; map indirection back to identity) in the same launch
__global__ void phd_import_kernel(float* __restrict__ slabs, int* __restrict__ counts, phd_pose* __restrict__ poses,
                                  const int* __restrict__ which, const unsigned char* __restrict__ buf, int cap,
                                  size_t stride, const int* __restrict__ rowsel, float* __restrict__ logw_fill, float nlw,
                                  int* __restrict__ parent_reset)
{
    const int k = blockIdx.x;
    const int p = which ? which[k] : k;
    const float* o = (const float*)(buf + (size_t)(rowsel ? rowsel[k] : k) * stride);
    if (threadIdx.x < 6) ((float*)&poses[p])[threadIdx.x] = o[threadIdx.x];
    if (threadIdx.x == 6) counts[p] = ((const int*)o)[6];
    if (threadIdx.x == 7) {
        if (logw_fill) logw_fill[p] = nlw;
        if (parent_reset) parent_reset[p] = p;
    }
    float* s = slabs + (size_t)p * 6 * cap;
    for (int i = threadIdx.x; i < 6 * cap; i += blockDim.x) s[i] = o[8 + i];
}

// The gathered multi-GPU resample in ONE launch (phd_global_resample_gathered, forced form): workgroup k of shard `off / n`
// runs the weights routine on the gathered row headers — every workgroup the whole normalise / fixed-point CDF, so there is
// no launch-to-launch dependency — draws the parent of ITS slot off + k, and copies that row into slot k
// (copy_particles, src/slamtypes.h:313-333).  The instantiation is the one launch_weights picks for n_global, so the index
// equals the single filter's bit for bit.  Workgroup 0 writes nEff / the decision / the normalised vector.
template <int BT, int R>
__global__ __launch_bounds__(BT) void phd_gathered_resample_kernel(WeightArgs A, float* __restrict__ slabs, int* __restrict__ counts,
                                                                   phd_pose* __restrict__ poses,
                                                                   const unsigned char* __restrict__ buf, int cap, size_t stride,
                                                                   int off, float* __restrict__ logw_fill, float nlw,
                                                                   int* __restrict__ parent_reset, float* __restrict__ cn_dst,
                                                                   int cn_len)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_gdyn[];
    __shared__ int s_row;
    const int k = blockIdx.x, tid = threadIdx.x;
    weights_body<BT, R, false, 1>(A, s_gdyn, off + k, k == 0, &s_row);
    __syncthreads();
    const float* o = (const float*)(buf + (size_t)s_row * stride);
    if (tid < 6) ((float*)&poses[k])[tid] = o[tid];
    if (tid == 6) counts[k] = ((const int*)o)[6];
    if (tid == 7) {
        if (logw_fill) logw_fill[k] = nlw;
        if (parent_reset) parent_reset[k] = k;
    }
    float* s = slabs + (size_t)k * 6 * cap;
    for (int i = tid; i < 6 * cap; i += BT) s[i] = o[8 + i];
    if (cn_dst)
        for (int i = tid; i < cn_len; i += BT) cn_dst[(size_t)k * cn_len + i] = o[8 + 6 * cap + i];
}

// copy_particles (src/slamtypes.h:313-333) after a global resample, both sources in ONE launch: slot j takes its parent from this
// shard (plan[j] >= 0: the local particle; map through the indirection `parent`) or from the received rows (plan[n + j]: the
// row of the all-to-all's receive buffer; a row may fill several slots).  Tail of copy_particles in the same launch:
// weights <- -log N (:327), the next step's map indirection <- identity (written to the OTHER indirection buffer: this
// launch still reads the current one).
__global__ void phd_resample_end_kernel(const float* __restrict__ src, const int* __restrict__ counts_src,
                                        const int* __restrict__ parent, const phd_pose* __restrict__ pose_src,
                                        const int* __restrict__ plan, int n, const unsigned char* __restrict__ recv, size_t stride,
                                        float* __restrict__ dst, int* __restrict__ counts_dst, phd_pose* __restrict__ pose_dst,
                                        int cap, float* __restrict__ logw_fill, float nlw, int* __restrict__ parent_next,
                                        const float* __restrict__ cn_src, float* __restrict__ cn_dst, int cn_len)
{
    const int j = blockIdx.x, tid = threadIdx.x;
    const int lp = plan[j];
    const float *a, *cn;
    int cnt;
    if (lp >= 0) {
        const int s = parent ? parent[lp] : lp;
        cnt = counts_src[s];
        a = src + (size_t)s * 6 * cap;
        cn = cn_src ? cn_src + (size_t)s * cn_len : nullptr;
        if (tid < 6) ((float*)&pose_dst[j])[tid] = ((const float*)&pose_src[lp])[tid];
    } else {
        const float* o = (const float*)(recv + (size_t)plan[n + j] * stride);
        cnt = ((const int*)o)[6];
        a = o + 8;
        cn = o + 8 + 6 * cap;
        if (tid < 6) ((float*)&pose_dst[j])[tid] = o[tid];
    }
    if (tid == 6) counts_dst[j] = cnt;
    if (tid == 7) {
        if (logw_fill) logw_fill[j] = nlw;
        if (parent_next) parent_next[j] = j;
    }
    float* b = dst + (size_t)j * 6 * cap;
    for (int pl = 0; pl < 6; ++pl)
        for (int i = tid; i < cnt; i += blockDim.x) b[pl * cap + i] = a[pl * cap + i];
    if (cn_dst)
        for (int i = tid; i < cn_len; i += blockDim.x) cn_dst[(size_t)j * cn_len + i] = cn[i];
}

// copy_particles after a global resample by direct reads of the owners' memory (phd_global_resample_pull): slot j of this
// shard takes global particle g = idx[off + j], i.e. particle g mod n of shard g / n, through that shard's view.  Systematic
// / stratified indices are non-decreasing, so the slots a remote parent fills are consecutive: phase 0 copies every local
// parent and the FIRST slot of every remote one (the only reads that cross the link), phase 1 fills the other slots of a
// remote parent from that first slot, which is local by then.  The tail of copy_particles rides along: weights <- -log N
// (:327), the next step's indirection <- identity.
struct PeerViews { phd_peer_view v[PHD_MAX_PEERS]; };
// (n: particles per OWNING shard — the divisor of the owner arithmetic; the grid is this shard's slot count)
__global__ void phd_resample_pull_kernel(PeerViews V, const int* __restrict__ idx, int off, int n, int rank, int phase,
                                         float* __restrict__ dst, int* __restrict__ counts_dst, phd_pose* __restrict__ pose_dst,
                                         int cap, float* __restrict__ logw_fill, float nlw, int* __restrict__ parent_next,
                                         float* __restrict__ cn_dst, int cn_len)
{
    const int j = blockIdx.x, tid = threadIdx.x;
    const int g = idx[off + j];
    const int owner = g / n, lp = g - owner * n;
    const bool remote = owner != rank;
    const bool first = j == 0 || idx[off + j - 1] != g;
    if (phase == 0 ? (remote && !first) : !(remote && !first)) return;
    float* b = dst + (size_t)j * 6 * cap;
    if (phase == 0) {
        const phd_peer_view& P = V.v[owner];
        const int s = P.parent ? P.parent[lp] : lp;
        const int cnt = P.counts[s];
        const float* a = P.maps + (size_t)s * 6 * cap;
        if (tid < 6) ((float*)&pose_dst[j])[tid] = ((const float*)&P.poses[lp])[tid];
        if (tid == 6) counts_dst[j] = cnt;
        for (int pl = 0; pl < 6; ++pl)
            for (int i = tid; i < cnt; i += blockDim.x) b[pl * cap + i] = a[pl * cap + i];
        if (cn_dst)
            for (int i = tid; i < cn_len; i += blockDim.x) cn_dst[(size_t)j * cn_len + i] = P.cn[(size_t)s * cn_len + i];
    } else {
        // the first slot this parent fills: lower bound of g in idx[off .. off + j)
        int lo = 0, hi = j;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (idx[off + mid] < g) lo = mid + 1; else hi = mid; }
        const int f0 = lo;
        const int cnt = counts_dst[f0];
        const float* a = dst + (size_t)f0 * 6 * cap;
        if (tid < 6) ((float*)&pose_dst[j])[tid] = ((const float*)&pose_dst[f0])[tid];
        if (tid == 6) counts_dst[j] = cnt;
        for (int pl = 0; pl < 6; ++pl)
            for (int i = tid; i < cnt; i += blockDim.x) b[pl * cap + i] = a[pl * cap + i];
        if (cn_dst)
            for (int i = tid; i < cn_len; i += blockDim.x) cn_dst[(size_t)j * cn_len + i] = cn_dst[(size_t)f0 * cn_len + i];
    }
    if (tid == 7) {
        if (logw_fill) logw_fill[j] = nlw;
        if (parent_next) parent_next[j] = j;
    }
}

// ------------------------------------------------------------------------------------------
// The COPY-FREE forms of the two exchanges above (round 5).  A single filter's resample moves no map: slot j's next update
// reads slab parent[j].  A shard's resample copied every slot — 32 us at 16 384 x 256, all of it LOCAL traffic — because a
// remote parent needs a slab of its own on this device.  The map buffers of a filter are ONE allocation [buffer 0 | buffer 1 |
// guests], n_max slabs each (phd_api.cpp), and the indirection counts slabs from the start of the CURRENT buffer, so it can
// name a guest: slab goff + j, goff = (2 - cur) n_max.  Local parents stay where they are (parent_next[j] = their slab, as in a
// single filter); the FIRST slot of a remote parent copies it into guest slab j — the only map traffic, and the only reads that
// cross the link — and the other slots that parent fills name the same guest.  The next update reads through the indirection
// and writes the other buffer; nothing is flipped here.  Counts and CPHD cardinality rows are laid out the same way.
// One wave per slot, four to a workgroup (most write two words and leave).  The callers use these forms only while the shard's
// indirection is the identity (no resample since the last update): a guest slab named by an older indirection must not be
// overwritten — the copying forms above take that case.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void copy_slab_wave(const float* __restrict__ a, int cnt, int cap, float* __restrict__ b, int tid, int nt)
{
    for (int pl = 0; pl < 6; ++pl)
        for (int i = tid; i < cnt; i += nt) b[pl * cap + i] = a[pl * cap + i];
}

__global__ void phd_resample_pull_free_kernel(PeerViews V, const int* __restrict__ idx, int off, int n, int rank,
                                              float* __restrict__ guests, int* __restrict__ counts_g, float* __restrict__ cn_g,
                                              int goff, phd_pose* __restrict__ pose_dst, int cap, float* __restrict__ logw_fill,
                                              float nlw, int* __restrict__ parent_next, int cn_len, const int* __restrict__ did,
                                              const float* __restrict__ logw_keep, int n_slots)
{
    // did (optional): the resample decision the weights launch left on the device (nEff trigger).  0: that launch wrote the
    // identity into idx, every slot "takes" its own particle below and its weight becomes this shard's slice of the normalised
    // vector (logw_keep) instead of -log N — the host, which enqueues this launch whatever the decision was, need not know it
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), tid = threadIdx.x & 63;   // one WAVE per slot
    if (j >= n_slots) return;
    if (did && !did[0]) nlw = logw_keep[j];
    const int g = idx[off + j];
    const int owner = g / n, lp = g - owner * n;
    const phd_peer_view& P = V.v[owner];
    if (tid < 6) ((float*)&pose_dst[j])[tid] = ((const float*)&P.poses[lp])[tid];
    const int s = P.parent ? P.parent[lp] : lp;          // the owner's slab of that particle (counted from ITS current buffer)
    if (owner == rank) {
        if (tid == 7) { if (parent_next) parent_next[j] = s; if (logw_fill) logw_fill[j] = nlw; }
        return;
    }
    const bool first = j == 0 || idx[off + j - 1] != g;
    if (!first) {
        if (tid == 7) {
            int lo = 0, hi = j;                              // the first slot this parent fills: lower bound of g in idx[off .. off + j)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (idx[off + mid] < g) lo = mid + 1; else hi = mid; }
            if (parent_next) parent_next[j] = goff + lo;
            if (logw_fill) logw_fill[j] = nlw;
        }
        return;
    }
    const int cnt = P.counts[s];
    copy_slab_wave(P.maps + (size_t)s * 6 * cap, cnt, cap, guests + (size_t)j * 6 * cap, tid, 64);
    if (cn_g)
        for (int i = tid; i < cn_len; i += 64) cn_g[(size_t)j * cn_len + i] = P.cn[(size_t)s * cn_len + i];
    if (tid == 6) counts_g[j] = cnt;
    if (tid == 7) { if (parent_next) parent_next[j] = goff + j; if (logw_fill) logw_fill[j] = nlw; }
}

// plan: [n] the local parent or -1 | [n] the received row | [n] the first slot that row fills
__global__ void phd_resample_end_free_kernel(const int* __restrict__ parent, const phd_pose* __restrict__ pose_src,
                                             const int* __restrict__ plan, int n, const unsigned char* __restrict__ recv, size_t stride,
                                             float* __restrict__ guests, int* __restrict__ counts_g, float* __restrict__ cn_g, int goff,
                                             phd_pose* __restrict__ pose_dst, int cap, float* __restrict__ logw_fill, float nlw,
                                             int* __restrict__ parent_next, int cn_len)
{
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), tid = threadIdx.x & 63;   // one WAVE per slot
    if (j >= n) return;
    const int lp = plan[j];
    if (lp >= 0) {
        if (tid < 6) ((float*)&pose_dst[j])[tid] = ((const float*)&pose_src[lp])[tid];
        if (tid == 7) { if (parent_next) parent_next[j] = parent ? parent[lp] : lp; if (logw_fill) logw_fill[j] = nlw; }
        return;
    }
    const float* o = (const float*)(recv + (size_t)plan[n + j] * stride);
    const int f0 = plan[2 * n + j];
    if (tid < 6) ((float*)&pose_dst[j])[tid] = o[tid];
    if (tid == 7) { if (parent_next) parent_next[j] = goff + f0; if (logw_fill) logw_fill[j] = nlw; }
    if (f0 != j) return;
    const int cnt = ((const int*)o)[6];
    copy_slab_wave(o + 8, cnt, cap, guests + (size_t)j * 6 * cap, tid, 64);
    if (cn_g)
        for (int i = tid; i < cn_len; i += 64) cn_g[(size_t)j * cn_len + i] = o[8 + 6 * cap + i];
    if (tid == 6) counts_g[j] = cnt;
}

// copy_particles for the maps (src/slamtypes.h:313-333): dst[p] = src[parent[sel[p]]] (maps, counts, poses);
// sel == NULL: identity; sel[p] < 0: slot is filled by phd_import_kernel instead
__global__ void phd_gather_maps_kernel(const float* __restrict__ src, const int* __restrict__ counts_src,
                                       const int* __restrict__ parent, const int* __restrict__ sel,
                                       float* __restrict__ dst, int* __restrict__ counts_dst,
                                       const phd_pose* __restrict__ pose_src, phd_pose* __restrict__ pose_dst, int cap)
{
    const int p = blockIdx.x;
    const int q = sel ? sel[p] : p;
    if (q < 0) return;
    const int s = parent ? parent[q] : q;
    const int n = counts_src[s];
    const float* a = src + (size_t)s * 6 * cap;
    float* b = dst + (size_t)p * 6 * cap;
    for (int i = threadIdx.x; i < 6 * cap; i += blockDim.x)
        if ((i % cap) < n) b[i] = a[i];
    if (threadIdx.x == 0) counts_dst[p] = n;
    if (pose_dst && threadIdx.x < 6) ((float*)&pose_dst[p])[threadIdx.x] = ((const float*)&pose_src[q])[threadIdx.x];
}

// rows of `len` floats (per-particle cardinality vectors): gather through up to two index maps, scatter through one
__global__ void phd_copy_rows_kernel(const float* __restrict__ src, size_t src_stride, const int* __restrict__ a,
                                     const int* __restrict__ b, float* __restrict__ dst, size_t dst_stride,
                                     const int* __restrict__ c, int len)
{
    const int k = blockIdx.x;
    int si = a ? a[k] : k;
    if (si < 0) return;
    if (b) si = b[si];
    const int di = c ? c[k] : k;
    const float* s = src + (size_t)si * src_stride;
    float* d = dst + (size_t)di * dst_stride;
    for (int i = threadIdx.x; i < len; i += blockDim.x) d[i] = s[i];
}

__global__ void phd_fill_kernel(float* a, float v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

__global__ void phd_iota_kernel(int* a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}

// ------------------------------------------------------------------------------------------
// launchers (called from phd_api.cpp; plain C++ signatures, no <<<>>> outside this file)
// ------------------------------------------------------------------------------------------
#define PHD_N_UPDATE_FNS_DECL 36
static const void* const k_update_fns[PHD_N_UPDATE_FNS_DECL] = {(const void*)phd_update_merge_kernel<false, false, false, false>,
                                             (const void*)phd_update_merge_kernel<true, false, false, false>,
                                             (const void*)phd_update_merge_kernel<false, true, false, false>,
                                             k_update_cphd_fns[0], k_update_cphd_fns[1], k_update_cphd_fns[2],
                                             (const void*)phd_update_merge_kernel<false, false, false, true>,
                                             (const void*)phd_update_merge_kernel<false, true, false, true>,
                                             k_update_cphd_fns[3], k_update_cphd_fns[4],
                                             k_update_w6_fns[0], k_update_w6_fns[1], k_update_w6_fns[2],
                                             k_update_cphd_w6_fns[0], k_update_cphd_w6_fns[1], k_update_cphd_w6_fns[2],
                                             k_update_w6_fns[3],
                                             // [17] the same block-form tail on the two-per-CU build (a layout that admits two workgroups per CU)
                                             (const void*)phd_update_merge_kernel<false, true, false, false, PHD_MIN_WAVES, true>,
                                             // [18..24] compiled-in layouts (the kernel template's LAYOUT): of [10], [12], [16] (three per CU, PHD), of [13],
                                             // [15] (three per CU, CPHD) with LAYOUT = 1, of [0], [2] (two per CU, PHD) with LAYOUT = 2
                                             k_update_w6_fns[4], k_update_w6_fns[5], k_update_w6_fns[6],
                                             k_update_cphd_w6_fns[3], k_update_cphd_w6_fns[4],
                                             (const void*)phd_update_merge_kernel<false, false, false, false, PHD_MIN_WAVES, false, 2>,
                                             (const void*)phd_update_merge_kernel<false, true, false, false, PHD_MIN_WAVES, false, 2>,
                                             // [25], [26]: the diagnostic instantiations (phase stamps) of the three-per-CU fast path, PHD and CPHD
                                             k_update_w6_fns[7], k_update_cphd_w6_fns[5],
                                             // [27..35] the layouts WITHOUT the scan length (LAYOUT = 3 / 4): of [18], [19], [20], [21], [22], [23], [24], [25], [26]
                                             k_update_w6_fns[8], k_update_w6_fns[9], k_update_w6_fns[10],
                                             k_update_cphd_w6_fns[6], k_update_cphd_w6_fns[7],
                                             (const void*)phd_update_merge_kernel<false, false, false, false, PHD_MIN_WAVES, false, 4>,
                                             (const void*)phd_update_merge_kernel<false, true, false, false, PHD_MIN_WAVES, false, 4>,
                                             k_update_w6_fns[11], k_update_cphd_w6_fns[8]};
#define PHD_N_UPDATE_FNS 36

// per-device one-time setup (function attributes are per device).  A mutex-guarded set of device ordinals: no aliasing of
// ordinals, no race between host threads that create or drive filters on different devices at the same time.
struct OncePerDevice {
    std::mutex mu;
    std::set<int> done;
};
static OncePerDevice g_update_attr;
template <typename F>
static hipError_t once_per_device(OncePerDevice& o, F&& setup)
{
    int dev_id = 0;
    hipError_t e = hipGetDevice(&dev_id);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(o.mu);
    if (o.done.count(dev_id)) return hipSuccess;
    e = setup();
    if (e == hipSuccess) o.done.insert(dev_id);
    return e;
}

// the largest static __shared__ footprint among the instantiations of the update kernel (the fused ones carry the
// weights routine's arrays): what a launch can use dynamically is 160 KiB minus this
size_t update_static_lds_bytes()
{
    size_t mx = 0;
    for (int k = 0; k < PHD_N_UPDATE_FNS; ++k) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, k_update_fns[k]) == hipSuccess && fa.sharedSizeBytes > mx) mx = fa.sharedSizeBytes;
    }
    return mx;
}

// index into k_update_fns: <STAMPS, FUSEW, CPHD, SPILL>, and the three-per-CU instantiations when three of this filter's
// workgroups fit the CU's LDS (the update's 608 B of static arrays included)
// compute units of the current device (cached per ordinal)
static int device_cu_count()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Does the runtime grant the 80-register instantiation `fn` three workgroups per CU at this LDS size?  (The arithmetic below says
// so for a CU with 160 KiB of LDS handed out in small granules; a device or driver that rounds differently would leave two — and
// the 80-register build without its reason.)  Asked once per device, function and size.
static bool three_granted(int fn, size_t lds_bytes)
{
    struct Key { int dev, fn; size_t lds; bool operator<(const Key& o) const { return dev != o.dev ? dev < o.dev : fn != o.fn ? fn < o.fn : lds < o.lds; } };
    static std::mutex mu;
    static std::map<Key, bool> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> g(mu);
    const Key k{dev, fn, lds_bytes};
    auto it = seen.find(k);
    if (it != seen.end()) return it->second;
    int n = 0;
    const bool ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_update_fns[fn], PHD_T, lds_bytes) == hipSuccess && n >= 3;
    seen[k] = ok;
    return ok;
}

// Which build a filter runs is decided ONCE, when the filter is created (phd_create stores the answer): the 80-register
// build pays ~2 % (and more on the critical path of a lone workgroup) for the right to a third resident workgroup, so it is taken
// only where LDS admits three, the runtime grants them AND the filter's base particle count has more than two workgroups per CU
// to place.  (Until round 4 this was re-derived from the launch's particle count on every launch: two hipGetDevice calls, a mutex
// and a map lookup per launch, and a filter whose count changes (the particle shotgun, phd_set_particle_count) could hop between
// builds.)  Call with the filter's device current.
bool update_takes_three_per_cu(bool cphd, bool spill, size_t lds_bytes, int n_base)
{
    if (spill || 3 * (lds_bytes + 1024) > 160 * 1024 || n_base <= 2 * device_cu_count()) return false;
    return three_granted(cphd ? 13 : 10, lds_bytes);
}

static int update_fn_index_any_layout(const UpdateArgs& a, bool three, int n_particles)
{
    const bool sp = a.spill_rec != nullptr && !a.stamps;           // (the diagnostic instantiation has no spill variant)
    const bool fused = a.fuse_weights && !a.stamps;
    // the fused step of more than 4096 particles: the instantiations with the block-form tail (PHD filters without a spill list: can_fuse)
    if (!sp && fused && !a.cphd && n_particles > PHD_GRID_WEIGHTS_MIN) return three ? 16 : 17;
    if (three && !sp) return (a.cphd ? 13 : 10) + (a.stamps ? 1 : fused ? 2 : 0);
    return a.stamps ? (a.cphd ? 4 : 1) : a.cphd ? (fused ? (sp ? 9 : 5) : (sp ? 8 : 3)) : (fused ? (sp ? 7 : 2) : (sp ? 6 : 0));
}
// ... and, where the filter's layout is one of the compiled-in ones and the scan is full, the instantiation that has both as constants (any_layout: never
// — a filter created with PHD_LAYOUT=0 in the environment: A/B, and the test that compares the two bit for bit)
static int update_fn_index(const UpdateArgs& a, bool three, int n_particles = 0, bool any_layout = false)
{
    const int fn = update_fn_index_any_layout(a, three, n_particles);
    if (any_layout) return fn;
    // a filter of a compiled-in layout: the instantiation with the scan's length too when the scan is full (18 ... 26), the one with the
    // layout alone otherwise (27 ... 35 = 9 further on)
    if (a.S_cap == FixedLayout<1>::S && a.cap == FixedLayout<1>::C && a.MM == FixedLayout<1>::MM) {
        const int dyn = (a.M == FixedLayout<1>::MM) ? 0 : 9;
        switch (fn) {
        case 10: case 12: case 16: if (a.cfg.distanceMetric == 0) return (fn == 10 ? 18 : fn == 12 ? 19 : 20) + dyn; break;   // (PHD: the Mahalanobis merge only)
        case 11: if (a.cfg.distanceMetric == 0) return 25 + dyn; break;
        case 14: if (a.cn_len == FixedLayout<1>::CN) return 26 + dyn; break;
        case 13: if (a.cn_len == FixedLayout<1>::CN) return 21 + dyn; break;       // (CPHD: the cardinality rows' length is compiled in as well)
        case 15: if (a.cn_len == FixedLayout<1>::CN) return 22 + dyn; break;
        default: break;
        }
    }
    if (a.S_cap == FixedLayout<2>::S && a.cap == FixedLayout<2>::C && a.MM == FixedLayout<2>::MM) {
        const int dyn = (a.M == FixedLayout<2>::MM) ? 0 : 9;
        if (a.cfg.distanceMetric == 0) switch (fn) { case 0: return 23 + dyn; case 2: return 24 + dyn; default: break; }
    }
    return fn;
}

int update_instantiation(const UpdateArgs& a, int n_particles, bool three_per_cu, bool any_layout)
{
    return update_fn_index(a, three_per_cu, n_particles, any_layout);
}

hipError_t launch_update_merge(const UpdateArgs& a, int n_particles, size_t lds_bytes, hipStream_t st, bool three_per_cu, bool any_layout)
{
    // function attributes are per device: set once for every device this process launches on (thread-safe: filters on
    // different devices may be driven by different host threads)
    {
        const hipError_t e = once_per_device(g_update_attr, [] {
            // dynamic LDS up to the CU's 160 KiB minus what the instantiation declares statically
            for (int k = 0; k < PHD_N_UPDATE_FNS; ++k) {
                hipFuncAttributes fa;
                hipError_t e = hipFuncGetAttributes(&fa, k_update_fns[k]);
                if (e != hipSuccess) return e;
                e = hipFuncSetAttribute(k_update_fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        });
        if (e != hipSuccess) return e;
    }
    const bool fused = a.fuse_weights && !a.stamps;
    const int fn = update_fn_index(a, three_per_cu, n_particles, any_layout);
    // + the weights workgroup(s) of the fused step: one up to 4096 particles, the block form's above
    const dim3 grid(n_particles + (fused ? (n_particles > PHD_GRID_WEIGHTS_MIN ? grid_weights_workgroups(n_particles) : 1) : 0)), b(PHD_T);
    UpdateArgs args = a;
    void* argv[] = {(void*)&args};
    return hipLaunchKernel(k_update_fns[fn], grid, b, argv, lds_bytes, st);
}

// workgroups of the update kernel a CU holds at a time for this filter (the runtime's own count: LDS, registers, waves)
int update_workgroups_per_cu(const UpdateArgs& a, size_t lds_bytes, bool three_per_cu, bool any_layout)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_update_fns[update_fn_index(a, three_per_cu, 0, any_layout)], PHD_T, lds_bytes) != hipSuccess) return -1;
    return n;
}

hipError_t launch_merge_spill(const UpdateArgs& a, int n_particles, hipStream_t st)
{
    if (!a.spill_rec || a.stamps) return hipSuccess;   // (the diagnostic instantiation has no spill variant)
    hipLaunchKernelGGL(phd_merge_spill_kernel, dim3(n_particles), dim3(PHD_T), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_copy_rows(const float* src, size_t src_stride, const int* a, const int* b, float* dst, size_t dst_stride,
                            const int* c, int len, int n, hipStream_t st)
{
    if (n <= 0 || len <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_copy_rows_kernel, dim3(n), dim3(256), 0, st, src, src_stride, a, b, dst, dst_stride, c, len);
    return hipGetLastError();
}

hipError_t launch_predict(const phd_pose* in, phd_pose* out, int n, phd_ackerman_control u,
                          const phd_ackerman_noise* noise, uint64_t seed, uint64_t counter, const DevConfig& cfg,
                          hipStream_t st)
{
    hipLaunchKernelGGL(phd_predict_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n, u, noise, seed,
                       counter, cfg);
    return hipGetLastError();
}

hipError_t launch_predict_shotgun(const phd_pose* in, phd_pose* out, int n_pred, int k, phd_ackerman_control u,
                                  const phd_ackerman_noise* noise, uint64_t seed, uint64_t counter, const DevConfig& cfg,
                                  const int* parent_in, int* parent_out, const float* logw_in, float* logw_out,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(phd_predict_shotgun_kernel, dim3((n_pred + 255) / 256), dim3(256), 0, st, in, out, n_pred, k, u, noise,
                       seed, counter, cfg, parent_in, parent_out, logw_in, logw_out);
    return hipGetLastError();
}

hipError_t launch_weights(const WeightArgs& a, hipStream_t st, unsigned* status)
{
    // above 4096 weights: the block form on several workgroups (phd_weights.h) - chosen by n ALONE, so that every caller and
    // every launch shape (this launch, the fused tail of the update kernel, every shard of a sharded filter) gets the same bits
    if (a.n > PHD_GRID_WEIGHTS_MIN && a.n_new <= a.n && a.gsync && a.gpart) {
        // its LDS is one block end per 256 weights: past 64 KB (about two million weights) the kernel needs the limit raised, and
        // past what a CU has (five million) the launch is refused instead of failing inside the runtime
        const size_t dyn_grid = grid_weights_lds_bytes(a.n);
        if (dyn_grid > 150 * 1024) return hipErrorInvalidValue;
        if (dyn_grid > 48 * 1024) {
            static OncePerDevice once_grid_attr;
            const hipError_t e = once_per_device(once_grid_attr, [] {
                return hipFuncSetAttribute((const void*)phd_weights_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            });
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(phd_weights_grid_kernel, dim3(grid_weights_workgroups(a.n)), dim3(PHD_T), dyn_grid, st, a, status);
        return hipGetLastError();
    }
    // up to 4096 weights: one workgroup, the fixed-point CDF in LDS; small particle sets use a small one (cheaper barriers, same
    // results: the reductions are fixed trees per block size)
    // (the commit of the small kernel writes logw[j] for j < n: it needs n_new <= n, true for every caller)
    const size_t dyn = (size_t)a.n * 8; // the fixed-point CDF, one u64 per particle
    // one table for this launcher and for the fused tail of the update kernel: 256 threads up
    // to 512 particles, 512 up to 4096; R = registers per thread.  (Round 4 also had 1024-thread forms with a 128 KB CDF for up to
    // 16 384 weights: replaced by the block form above.)
    const bool small = a.n_new <= a.n && a.n <= PHD_GRID_WEIGHTS_MIN;
    if (small) {
        WeightArgs b = a;
        // a pure index draw leaves the weights as they are: no write-back (and no alias for the split kernel to mind)
        if (b.logw == b.logw_in && !(b.mode & (W_ACCUMULATE | W_NORMALIZE | W_COMMIT))) b.logw = nullptr;
        const bool split = (b.mode & (W_RESAMPLE_FORCE | W_RESAMPLE_AUTO)) && b.n > 1024 && b.logw != b.logw_in;
        // (above 1024 particles the index searches and copy_particles are split over 8 workgroups: same bits, a third of the time)
        if (split) hipLaunchKernelGGL((phd_weights_split_kernel<512, 8, 8>), dim3(8), dim3(512), dyn, st, b);
        else if (b.n <= 256) hipLaunchKernelGGL((phd_weights_small_kernel<256, 1>), dim3(1), dim3(256), dyn, st, b);
        else if (b.n <= 512) hipLaunchKernelGGL((phd_weights_small_kernel<256, 2>), dim3(1), dim3(256), dyn, st, b);
        else if (b.n <= 1024) hipLaunchKernelGGL((phd_weights_small_kernel<512, 2>), dim3(1), dim3(512), dyn, st, b);
        else hipLaunchKernelGGL((phd_weights_small_kernel<512, 8>), dim3(1), dim3(512), dyn, st, b);
        return hipGetLastError();
    }
    // (a resample that GROWS the set, or a caller without the block form's scratch: the general one-workgroup kernel, CDF in HBM)
    if (a.n <= 1024) hipLaunchKernelGGL(phd_weights_kernel<256>, dim3(1), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(phd_weights_kernel<1024>, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_pack_maps(const phd_gaussian2d* concat, const int* offsets, const int* sizes, float* slabs, int cap,
                            int n, hipStream_t st)
{
    hipLaunchKernelGGL(phd_pack_maps_kernel, dim3(n), dim3(128), 0, st, concat, offsets, sizes, slabs, cap);
    return hipGetLastError();
}

hipError_t launch_unpack_maps(const float* slabs, const int* parent, const int* offsets, const int* counts,
                              phd_gaussian2d* concat, int cap, int n, hipStream_t st)
{
    hipLaunchKernelGGL(phd_unpack_maps_kernel, dim3(n), dim3(128), 0, st, slabs, parent, offsets, counts, concat, cap);
    return hipGetLastError();
}

// one particle's map, the particle index taken from device memory (the arg-max of phd_state_kernel): AoS out, count out
__global__ void phd_unpack_one_kernel(const float* __restrict__ slabs, const int* __restrict__ parent,
                                      const int* __restrict__ counts, const int* __restrict__ which,
                                      phd_gaussian2d* __restrict__ out, int cap, int* __restrict__ n_out)
{
    const int p = which[0];
    if (p < 0) { if (threadIdx.x == 0) n_out[0] = 0; return; }
    const int src = parent ? parent[p] : p;
    const int n = counts[src];
    const float* s = slabs + (size_t)src * 6 * cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        phd_gaussian2d v;
        v.weight = s[0 * cap + i];
        v.mean[0] = s[1 * cap + i];
        v.mean[1] = s[2 * cap + i];
        v.cov[0] = s[3 * cap + i];
        v.cov[1] = s[4 * cap + i];
        v.cov[2] = s[4 * cap + i];
        v.cov[3] = s[5 * cap + i];
        out[i] = v;
    }
    if (threadIdx.x == 0) n_out[0] = n;
}

hipError_t launch_unpack_one(const float* slabs, const int* parent, const int* counts, const int* which, phd_gaussian2d* out,
                             int cap, int* n_out, hipStream_t st)
{
    hipLaunchKernelGGL(phd_unpack_one_kernel, dim3(1), dim3(256), 0, st, slabs, parent, counts, which, out, cap, n_out);
    return hipGetLastError();
}

hipError_t launch_state(const phd_pose* poses, const float* logw, int n, float* pose_out, int* argmax_out,
                        hipStream_t st)
{
    hipLaunchKernelGGL(phd_state_kernel, dim3(1), dim3(PHD_WT), 0, st, poses, logw, n, pose_out, argmax_out);
    return hipGetLastError();
}

hipError_t launch_export(const float* slabs, const int* counts, const int* parent, const phd_pose* poses,
                         const int* which, void* buf, int cap, size_t stride, int n, hipStream_t st, const float* raw)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_export_kernel, dim3(n), dim3(256), 0, st, slabs, counts, parent, poses, which,
                       (unsigned char*)buf, cap, stride, raw);
    return hipGetLastError();
}

hipError_t launch_import(float* slabs, int* counts, phd_pose* poses, const int* which, const void* buf, int cap,
                         size_t stride, int n, hipStream_t st, const int* rowsel, float* logw_fill, float nlw,
                         int* parent_reset)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_import_kernel, dim3(n), dim3(256), 0, st, slabs, counts, poses, which,
                       (const unsigned char*)buf, cap, stride, rowsel, logw_fill, nlw, parent_reset);
    return hipGetLastError();
}

// n_global <= 1024 only (above, n workgroups repeating the scan cost more than the launch they save)
hipError_t launch_gathered_resample(const WeightArgs& a, float* slabs, int* counts, phd_pose* poses, const void* rows, int cap,
                                    size_t stride, int off, int n, float* logw_fill, float nlw, int* parent_reset,
                                    float* cn_dst, int cn_len, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (a.n > 1024 || a.n_new != a.n) return hipErrorInvalidValue;
    const size_t dyn = (size_t)a.n * 8;
#define PHD_GR(BT, R)                                                                                                       \
    hipLaunchKernelGGL((phd_gathered_resample_kernel<BT, R>), dim3(n), dim3(BT), dyn, st, a, slabs, counts, poses,            \
                       (const unsigned char*)rows, cap, stride, off, logw_fill, nlw, parent_reset, cn_dst, cn_len)
    if (a.n <= 256) PHD_GR(256, 1);       // the table of launch_weights
    else if (a.n <= 512) PHD_GR(256, 2);
    else PHD_GR(512, 2);
#undef PHD_GR
    return hipGetLastError();
}

hipError_t launch_resample_end(const float* src, const int* counts_src, const int* parent, const phd_pose* pose_src, const int* plan,
                               int n, const void* recv, size_t stride, float* dst, int* counts_dst, phd_pose* pose_dst, int cap,
                               float* logw_fill, float nlw, int* parent_next, const float* cn_src, float* cn_dst, int cn_len,
                               hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_resample_end_kernel, dim3(n), dim3(256), 0, st, src, counts_src, parent, pose_src, plan, n,
                       (const unsigned char*)recv, stride, dst, counts_dst, pose_dst, cap, logw_fill, nlw, parent_next, cn_src,
                       cn_dst, cn_len);
    return hipGetLastError();
}

hipError_t launch_resample_pull(const phd_peer_view* views, int world, const int* idx, int off, int n_src, int n_dst, int rank, float* dst,
                                int* counts_dst, phd_pose* pose_dst, int cap, float* logw_fill, float nlw, int* parent_next,
                                float* cn_dst, int cn_len, hipStream_t st)
{
    if (n_dst <= 0) return hipSuccess;
    if (world > PHD_MAX_PEERS) return hipErrorInvalidValue;
    PeerViews V = {};
    for (int k = 0; k < world; ++k) V.v[k] = views[k];
    for (int phase = 0; phase < (world > 1 ? 2 : 1); ++phase)
        hipLaunchKernelGGL(phd_resample_pull_kernel, dim3(n_dst), dim3(256), 0, st, V, idx, off, n_src, rank, phase, dst, counts_dst, pose_dst,
                           cap, logw_fill, nlw, parent_next, cn_dst, cn_len);
    return hipGetLastError();
}

hipError_t launch_resample_pull_free(const phd_peer_view* views, int world, const int* idx, int off, int n_src, int n_dst, int rank,
                                     float* guests, int* counts_g, float* cn_g, int goff, phd_pose* pose_dst, int cap, float* logw_fill,
                                     float nlw, int* parent_next, int cn_len, hipStream_t st, const int* did, const float* logw_keep)
{
    if (n_dst <= 0) return hipSuccess;
    if (world > PHD_MAX_PEERS) return hipErrorInvalidValue;
    PeerViews V = {};
    for (int k = 0; k < world; ++k) V.v[k] = views[k];
    hipLaunchKernelGGL(phd_resample_pull_free_kernel, dim3((n_dst + 3) / 4), dim3(256), 0, st, V, idx, off, n_src, rank, guests, counts_g, cn_g,
                       goff, pose_dst, cap, logw_fill, nlw, parent_next, cn_len, did, logw_keep, n_dst);
    return hipGetLastError();
}

hipError_t launch_resample_end_free(const int* parent, const phd_pose* pose_src, const int* plan, int n, const void* recv, size_t stride,
                                    float* guests, int* counts_g, float* cn_g, int goff, phd_pose* pose_dst, int cap, float* logw_fill,
                                    float nlw, int* parent_next, int cn_len, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_resample_end_free_kernel, dim3((n + 3) / 4), dim3(256), 0, st, parent, pose_src, plan, n, (const unsigned char*)recv, stride,
                       guests, counts_g, cn_g, goff, pose_dst, cap, logw_fill, nlw, parent_next, cn_len);
    return hipGetLastError();
}

hipError_t launch_gather_maps(const float* src, const int* counts_src, const int* parent, const int* sel, float* dst,
                              int* counts_dst, const phd_pose* pose_src, phd_pose* pose_dst, int cap, int n,
                              hipStream_t st)
{
    hipLaunchKernelGGL(phd_gather_maps_kernel, dim3(n), dim3(256), 0, st, src, counts_src, parent, sel, dst, counts_dst,
                       pose_src, pose_dst, cap);
    return hipGetLastError();
}

hipError_t launch_fill(float* a, float v, int n, hipStream_t st)
{
    hipLaunchKernelGGL(phd_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, v, n);
    return hipGetLastError();
}

hipError_t launch_iota(int* a, int n, hipStream_t st)
{
    hipLaunchKernelGGL(phd_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
    return hipGetLastError();
}

#endif // !PHD_PART_TU
} // namespace phd
