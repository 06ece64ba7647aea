// phd_weights.h — particle weights: accumulate, logSumExp normalise, nEff, fixed-point CDF resample (one workgroup).
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"
#include "phd_sort.h"
#include "phd_merge.h"
#include "phd_predict.h"
#include "phd_cphd.h"

namespace phd {

// ------------------------------------------------------------------------------------------
// particle weights: accumulate, logSumExp normalise, nEff, resample (one workgroup)
// ------------------------------------------------------------------------------------------
// portable exp for the resampling CDF: IEEE basic operations only (mul, fma, rint, ldexp), so
// the double it returns is the same on every conforming CPU and GPU (see oracle/scphd_cpu.c).
// -> det_exp() in phd_detexp.h (shared with phd_eap.hip)

// ------------------------------------------------------------------------------------------
// Fixed-point resampling CDF (definition and rationale: oracle/scphd_cpu.c, o_resample):
//   sb = 62 - ceil(log2 N);  q_i = floor(min(det_exp(w_i), 1) * 2^sb);  Q_i = q_0 + ... + q_i (exact)
//   the reference's "r_j > c_i"  <=>  Q_i < T_j = ceil(r_j * 2^sb)
// Integer sums are associative: the parallel scan below, the oracle's sequential loop and every rank
// of a multi-GPU run produce the same Q, hence the same indices.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int cdf_scale_bits(int n)
{
    int b = 0;
    while ((1ll << b) < n) ++b;
    return 62 - b;
}
__device__ __forceinline__ u64 cdf_quantise(double p, double scale) { return (u64)floor((p > 1.0 ? 1.0 : p) * scale); }

// in-place inclusive scan of q[0..m) (u64, LDS) by a workgroup of BT threads, plus `carry`; returns
// the total (carry included).  Wave w owns the contiguous block [w*64*per, (w+1)*64*per) and walks it in rows of 64
// consecutive entries (lane l of a row reads entry row*64 + l: consecutive lanes, consecutive banks — a thread that owned
// `per` consecutive entries would put the lanes 8*per bytes apart, a 32-way bank conflict at 16 per thread, which made this
// scan 14.7 us at 16384 particles).  Integer sums: any grouping gives the same Q.
__device__ __forceinline__ u64 lane63_u64(u64 v)
{
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, 63), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), 63);
    return ((u64)hi << 32) | lo;
}
template <int BT>
__device__ __forceinline__ u64 block_scan_u64(u64* q, int m, u64 carry, u64* s_wtot, int tid)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const int per = (m + BT - 1) / BT;
    const int base = wave * 64 * per + lane;
    u64 local = 0;
    for (int e = 0; e < per; ++e) {
        const int i = base + e * 64;
        if (i < m) local += q[i];
    }
    const u64 wsum = lane63_u64(wave_incl_scan(local));
    if (lane == 63) s_wtot[wave] = wsum;
    __syncthreads();
    u64 run = carry, total = carry;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) {
        const u64 c = s_wtot[w];
        if (w < wave) run += c;
        total += c;
    }
    for (int e = 0; e < per; ++e) {
        const int i = base + e * 64;
        const u64 incl = wave_incl_scan(i < m ? q[i] : 0ull);
        if (i < m) q[i] = run + incl;
        run += lane63_u64(incl);
    }
    __syncthreads();
    return total;
}

#define PHD_CDF_CHUNK 2048

template <int PHD_WT>
__device__ __forceinline__ float block_reduce_w(float v, float* sc, int tid, bool is_max)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float o = xor_lane(v, off);
        v = is_max ? fmaxf(v, o) : (v + o);
    }
    __syncthreads();
    if ((tid & 63) == 0) sc[tid >> 6] = v;
    __syncthreads();
    float r = sc[0];
    for (int w = 1; w < PHD_WT / 64; ++w) r = is_max ? fmaxf(r, sc[w]) : (r + sc[w]);
    return r;
}

// mode bits
enum { W_ACCUMULATE = 1, W_NORMALIZE = 2, W_RESAMPLE_FORCE = 4, W_RESAMPLE_AUTO = 8, W_HAD_MEAS = 16, W_COMMIT = 32 };

template <int PHD_WT>
__global__ __launch_bounds__(PHD_WT) void phd_weights_kernel(WeightArgs A)
{
    __shared__ float sc[PHD_WT / 64];
    __shared__ int s_flag;
    __shared__ int s_argmax;
    __shared__ double s_chunk[PHD_CDF_CHUNK];
    __shared__ u64 s_wtot[PHD_WT / 64];
    __shared__ double s_bestv[PHD_WT / 64];
    __shared__ int s_besti[PHD_WT / 64];
    const int tid = threadIdx.x;
    const int n = A.n;          // weights in the vector being normalised (global count for multi-GPU)
    float* logw = A.logw;       // [n] working / output vector (== logw_in unless the filter is frozen)
    // 1. accumulate the increments of the last update (src/phdfilter.cu:3741-3744)
    if ((A.mode & W_ACCUMULATE) || A.logw_in != logw) {
        const size_t ls = A.in_stride ? (size_t)A.in_stride : 1;
        for (int i = tid; i < n; i += PHD_WT) {
            float w = A.logw_in[i * ls];
            if (A.mode & W_ACCUMULATE) w += A.dlogw[i];
            logw[i] = w;
            if (A.raw_out) A.raw_out[i] = w;
        }
        __syncthreads();
    }
    // 2. logSumExp normalise (src/device_math.cuh:549-558, src/phdfilter.cu:3749-3754)
    if (A.mode & W_NORMALIZE) {
        float mx = -FLT_MAX;
        for (int i = tid; i < n; i += PHD_WT) mx = fmaxf(mx, logw[i]);
        mx = block_reduce_w<PHD_WT>(mx, sc, tid, true);
        float s = 0.f;
        for (int i = tid; i < n; i += PHD_WT) s += expf(logw[i] - mx);
        s = block_reduce_w<PHD_WT>(s, sc, tid, false);
        const float lse = safe_log(s) + mx;
        for (int i = tid; i < n; i += PHD_WT) logw[i] -= lse;
        __syncthreads();
    }
    // 3. nEff = 1 / sum exp(2w) / N (src/main.cpp:1281-1284)
    float s2 = 0.f;
    for (int i = tid; i < n; i += PHD_WT) s2 += expf(2 * logw[i]);
    s2 = block_reduce_w<PHD_WT>(s2, sc, tid, false);
    const float neff = (float)(1.0 / (double)s2 / (double)n);
    if (tid == 0) {
        A.neff_out[0] = neff;
        int doit = 0;
        if (A.mode & W_RESAMPLE_FORCE) doit = 1;
        else if ((A.mode & W_RESAMPLE_AUTO) && (neff <= A.resample_thresh) && (A.mode & W_HAD_MEAS)) doit = 1; // :1286
        s_flag = doit;
        A.did_resample[0] = doit;
    }
    __syncthreads();
    const int n_new = A.n_new;
    if (!s_flag) {
        for (int j = tid; j < ((A.mode & W_COMMIT) ? n : n_new); j += PHD_WT) {
            A.idx_out[j] = j;                                                                          // :1292-1296
            if (A.mode & W_COMMIT) {
                A.pose_out[j] = A.pose_in[j];
                A.parent_out[j] = A.parent_in[j];
            }
        }
        return;
    }
    // 4. resample (src/main.cpp:453-501).  Thresholds: HEAD's expression r_j = j*interval + u_j*interval
    //    (:468); with a single uniform (systematic, as src/phdfilter.cu.bak:3279-3327) u_j = u_0.
    //    CDF in fixed point (see cdf_quantise): chunks of 2048 scanned in LDS, spilled to A.cdf (as u64).
    u64* cdf = (u64*)A.cdf;   // [n] global scratch
    u64* qch = (u64*)s_chunk;
    const double interval = 1.0 / n_new;
    const int sb = cdf_scale_bits(n);
    const double scale = ldexp(1.0, sb);
    double best = -1.0;
    int besti = 0x7FFFFFFF;
    u64 carry = 0;
    for (int c0 = 0; c0 < n; c0 += PHD_CDF_CHUNK) {
        const int m = (n - c0 < PHD_CDF_CHUNK) ? (n - c0) : PHD_CDF_CHUNK;
        for (int i = tid; i < m; i += PHD_WT) {
            const double e = det_exp(logw[c0 + i]);
            qch[i] = cdf_quantise(e, scale);
            if (e > best) { best = e; besti = c0 + i; } // strided ascending: keeps the lowest index per lane
        }
        __syncthreads();
        carry = block_scan_u64<PHD_WT>(qch, m, carry, s_wtot, tid);
        for (int i = tid; i < m; i += PHD_WT) cdf[c0 + i] = qch[i];
        __syncthreads();
    }
    // arg-max of p (first maximum, strict '>'), used by the overflow guard (:475-494)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = xor_lane(best, off);
            const int oi = xor_lane(besti, off);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if ((tid & 63) == 0) { s_bestv[tid >> 6] = best; s_besti[tid >> 6] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < PHD_WT / 64; ++w)
                if (s_bestv[w] > best || (s_bestv[w] == best && s_besti[w] < besti)) { best = s_bestv[w]; besti = s_besti[w]; }
            s_argmax = besti;
        }
        __syncthreads();
    }
    const u64 ctot = carry;
    for (int j = tid; j < n_new; j += PHD_WT) {
        const double u = (A.n_uniforms == 1) ? A.u0 : A.uniforms[j];
        const double r = j * interval + u * interval;                                                  // :468
        const u64 T = (u64)ceil(r * scale);
        int idx;
        if (T > ctot) {
            idx = s_argmax;                                                                            // :475-494
        } else {
            // smallest i with Q_i >= T  ==  where the reference's "while (r > c) i++" stops
            int lo = 0, hi = n - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] < T) lo = mid + 1; else hi = mid;
            }
            idx = lo;
        }
        A.idx_out[j] = idx;
    }
    if (A.mode & W_COMMIT) {
        // copy_particles (src/slamtypes.h:313-333): gather poses, compose the map indirection,
        // weights <- -log(N)
        const float nlw = (float)(-log((double)A.n_weight_norm));
        __syncthreads();
        for (int j = tid; j < n_new; j += PHD_WT) {
            const int s = A.idx_out[j];
            A.pose_out[j] = A.pose_in[s];
            A.parent_out[j] = A.parent_in[s];
            logw[j] = nlw;
        }
    }
}

// ------------------------------------------------------------------------------------------
// the same routine for n <= BT*R with the weights held in registers from load to commit: one
// global read of (logw, dlogw), three block reductions, the sequential CDF in LDS, one global
// write.  n <= PHD_CDF_CHUNK.  Results are a pure function of (inputs, BT): every rank of a
// multi-GPU run launches the same instantiation on the same gathered vector.
// ------------------------------------------------------------------------------------------
// hand-off loads (fused step): data written by OTHER workgroups of the same launch is read with
// agent-scope (sc1) loads, which bypass this CU's L1 (cdna guide, Guideline 16)
template <bool HANDOFF>
__device__ __forceinline__ float ld_f32(const float* p)
{
    return HANDOFF ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool HANDOFF>
__device__ __forceinline__ int ld_i32(const int* p)
{
    return HANDOFF ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool HANDOFF>
__device__ __forceinline__ phd_pose ld_pose(const phd_pose* p)
{
    if (!HANDOFF) return *p;
    const float* f = (const float*)p;
    phd_pose o;
    o.px = ld_f32<true>(f + 0); o.py = ld_f32<true>(f + 1); o.ptheta = ld_f32<true>(f + 2);
    o.vx = ld_f32<true>(f + 3); o.vy = ld_f32<true>(f + 4); o.vtheta = ld_f32<true>(f + 5);
    return o;
}

// WIN > 0 (slot window): the workgroup runs the whole routine on the weights but draws (and commits) only the resampling
// slots [slot, slot + WIN * BT) — WIN searches per thread; only the `lead` workgroup writes the shared outputs (nEff,
// decision, normalised weights).  Every workgroup of such a launch computes the same normalisation and the same CDF, so
// there is nothing to exchange between them.  Two users: the gathered multi-GPU resample (WIN = 1, one slot per import
// workgroup, its parent left in *one_out, LDS, valid after the caller's barrier) and phd_weights_split_kernel (the
// searches and copy_particles of a large particle set split over several CUs).
template <int BT, int R, bool HANDOFF, int WIN = 0>
__device__ __forceinline__ void weights_body(const WeightArgs& A, unsigned char* s_dyn, int slot = 0, bool lead = true,
                                             int* one_out = nullptr)
{
    constexpr bool WINDOWED = WIN > 0;
    __shared__ float sc[BT / 64];
    __shared__ int s_argmax;
    __shared__ u64 s_wtot[BT / 64];
    __shared__ double s_bestv[BT / 64];
    __shared__ int s_besti[BT / 64];
    const int tid = threadIdx.x;
    const int n = A.n;
#define WSTAMP(k) do { if (A.wstamps && tid == 0 && (!WINDOWED || lead)) A.wstamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
    WSTAMP(0);
    float w[R];
    // 1. load + accumulate (src/phdfilter.cu:3741-3744)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * BT;
        w[r] = -FLT_MAX;
        if (i < n) {
            w[r] = A.logw_in[A.in_stride ? (size_t)i * A.in_stride : (size_t)i];
            if (A.mode & W_ACCUMULATE) w[r] += ld_f32<HANDOFF>(&A.dlogw[i]);
            if (A.raw_out && (!WINDOWED || lead)) A.raw_out[i] = w[r];
        }
    }
    // 2. logSumExp normalise (src/device_math.cuh:549-558, src/phdfilter.cu:3749-3754)
    if (A.mode & W_NORMALIZE) {
        float mx = -FLT_MAX;
#pragma unroll
        for (int r = 0; r < R; ++r) mx = fmaxf(mx, w[r]);
        mx = block_reduce_w<BT>(mx, sc, tid, true);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (tid + r * BT < n) s += expf(w[r] - mx);
        s = block_reduce_w<BT>(s, sc, tid, false);
        const float lse = safe_log(s) + mx;
#pragma unroll
        for (int r = 0; r < R; ++r) w[r] -= lse;
    }
    WSTAMP(1);
    // 3. nEff (src/main.cpp:1281-1284)
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) if (tid + r * BT < n) s2 += expf(2 * w[r]);
    s2 = block_reduce_w<BT>(s2, sc, tid, false);
    const float neff = (float)(1.0 / (double)s2 / (double)n);
    int doit = 0;
    if (A.mode & W_RESAMPLE_FORCE) doit = 1;
    else if ((A.mode & W_RESAMPLE_AUTO) && (neff <= A.resample_thresh) && (A.mode & W_HAD_MEAS)) doit = 1; // :1286
    if (tid == 0 && (!WINDOWED || lead)) { A.neff_out[0] = neff; A.did_resample[0] = doit; }
    const int n_new = A.n_new;
    if (!doit) { // uniform: neff is the same in every thread
        if (A.logw && (!WINDOWED || lead)) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = tid + r * BT;
                if (i < n) A.logw[i] = w[r];
            }
        }
        if (WINDOWED) { // (a window launch is always a forced resample; kept total for completeness)
            const int lim = (A.mode & W_COMMIT) ? n : n_new;
            const int sp = one_out ? 1 : WIN * BT;
            const int je = (slot + sp < lim) ? slot + sp : lim;
            for (int j = slot + tid; j < je; j += BT) {
                A.idx_out[j] = j;
                if (A.mode & W_COMMIT) { A.pose_out[j] = ld_pose<HANDOFF>(&A.pose_in[j]); A.parent_out[j] = ld_i32<HANDOFF>(&A.parent_in[j]); }
            }
            if (one_out && tid == 0) *one_out = slot;
            return;
        }
        for (int j = tid; j < ((A.mode & W_COMMIT) ? n : n_new); j += BT) {
            A.idx_out[j] = j;                                                                          // :1292-1296
            if (A.mode & W_COMMIT) { A.pose_out[j] = ld_pose<HANDOFF>(&A.pose_in[j]); A.parent_out[j] = ld_i32<HANDOFF>(&A.parent_in[j]); }
        }
        return;
    }
    // 4. resample: p_i = det_exp(w_i) -> fixed-point CDF (see cdf_quantise) scanned in LDS by the
    //    whole workgroup; thresholds r_j = j*interval + u*interval (src/main.cpp:468)
    WSTAMP(2);
    u64* Q = (u64*)s_dyn; // [n]
    const int sb = cdf_scale_bits(n);
    const double scale = ldexp(1.0, sb);
    double best = -1.0;
    int besti = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * BT;
        if (i < n) {
            const double e = det_exp(w[r]);
            Q[i] = cdf_quantise(e, scale);
            if (e > best) { best = e; besti = i; }
        }
    }
    __syncthreads();
    WSTAMP(3);
    const u64 ctot = block_scan_u64<BT>(Q, n, 0ull, s_wtot, tid);
    WSTAMP(4);
    const double interval = 1.0 / n_new;
    // the overflow guard (src/main.cpp:475-494) needs the arg-max of p only if the last threshold
    // exceeds the total mass (weights that do not sum to one): thresholds increase with j
    {
        const int jl = n_new - 1;
        const double ul = (A.n_uniforms == 1) ? A.u0 : A.uniforms[jl];
        const bool overflow = (u64)ceil((jl * interval + ul * interval) * scale) > ctot;
        if (overflow) { // uniform
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = xor_lane(best, off);
                const int oi = xor_lane(besti, off);
                if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
            }
            if ((tid & 63) == 0) { s_bestv[tid >> 6] = best; s_besti[tid >> 6] = besti; }
            __syncthreads();
            if (tid == 0) {
                for (int wv = 1; wv < BT / 64; ++wv)
                    if (s_bestv[wv] > best || (s_bestv[wv] == best && s_besti[wv] < besti)) { best = s_bestv[wv]; besti = s_besti[wv]; }
                s_argmax = besti;
            }
            __syncthreads();
        }
    }
    WSTAMP(5);
    const float nlw = (float)(-log((double)A.n_weight_norm));
    // smallest i with Q_i >= T_j  ==  where the reference's "while (r > c) i++" stops (src/main.cpp:470-473).  The thread's
    // R searches advance together, one power-of-two step per trip (branch-free lower bound): R independent LDS reads in
    // flight per level instead of R x log2(n) dependent ones — the searches were half of this routine's time at 4096
    // particles, which the fused step spends with every other workgroup already gone.
    constexpr int RS = WINDOWED ? WIN : R;
    const int j_first = WINDOWED ? slot : 0;
    const int span = one_out ? 1 : WIN * BT;   // the gathered resample draws ONE slot per workgroup
    const int j_end = (WINDOWED && slot + span < n_new) ? slot + span : n_new;
    u64 T[RS];
    int pos[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int jj = j_first + tid + r * BT;
        const double u = (A.n_uniforms == 1) ? A.u0 : A.uniforms[jj < j_end ? jj : 0];
        const double rr = jj * interval + u * interval;                                                // :468
        T[r] = jj < j_end ? (u64)ceil(rr * scale) : 0ull; // 0: no Q is below it, the search stays at 0
        pos[r] = 0;
    }
    for (int step = 1 << (31 - __clz(n)); step > 0; step >>= 1) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int probe = pos[r] + step; // pos = number of entries known to be < T
            const u64 qv = Q[(probe <= n ? probe : n) - 1];
            if (probe <= n && qv < T[r]) pos[r] = probe;
        }
    }
    WSTAMP(7);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int jj = j_first + tid + r * BT;
        if (jj >= j_end) continue;
        const int idx = (T[r] > ctot) ? s_argmax : pos[r];                                             // :475-494
        A.idx_out[jj] = idx;
        if (WINDOWED && one_out && jj == slot) *one_out = idx;
        if (A.mode & W_COMMIT) { // copy_particles (src/slamtypes.h:313-333)
            A.pose_out[jj] = ld_pose<HANDOFF>(&A.pose_in[idx]);
            A.parent_out[jj] = ld_i32<HANDOFF>(&A.parent_in[idx]);
        }
    }
    // weights: -log(N) after a committed resample (slamtypes.h:327), else the normalised values
    if (A.logw && (!WINDOWED || lead)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * BT;
            if (i < n) A.logw[i] = (A.mode & W_COMMIT) ? nlw : w[r];
        }
    }
    WSTAMP(6);
}

// Large particle sets: the searches and copy_particles (half of the routine, instruction-bound on one CU) split over K
// workgroups of slot windows; the normalisation and the CDF are repeated by each (identical: same block size, same trees).
// The workgroups do not wait for each other, so the lead's weight output must not alias anybody's input: launch_weights
// uses this kernel only when A.logw is NULL or a buffer other than A.logw_in.
template <int BT, int R, int K>
__global__ __launch_bounds__(BT) void phd_weights_split_kernel(WeightArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_wdyn[];
    static_assert(R % K == 0, "slot windows");
    weights_body<BT, R, false, R / K>(A, s_wdyn, (int)blockIdx.x * (R / K) * BT, blockIdx.x == 0, nullptr);
}

template <int BT, int R>
__global__ __launch_bounds__(BT) void phd_weights_small_kernel(WeightArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_wdyn[];
    weights_body<BT, R, false>(A, s_wdyn);
}


// ------------------------------------------------------------------------------------------
// Large particle sets (n > PHD_GRID_WEIGHTS_MIN): the routine on SEVERAL workgroups, no CDF in LDS (round 5).
//
// The single-workgroup forms above keep the fixed-point CDF of all n weights in LDS (128 KB at 16 384 particles: a whole CU,
// 27-31 us, a second launch after the update kernel, and on a sharded filter pure serial tail repeated by every shard).  Here the
// vector is cut into BLOCKS of 256 consecutive weights; a workgroup of 512 threads takes two blocks per pass, one weight per
// thread, and the workgroups meet at two grid-wide barriers (an arrival counter in HBM; data crosses with sc1 stores, a drained
// vmcnt and sc1 loads - the hand-off of the fused step, cdna guide Guideline 16):
//
//   pass A   x_i = logw_i (+ dlogw_i);  per block  m_b = max x_i,  s_b = sum expf(x_i - m_b)            -> part[b]
//   -------- barrier --------
//   combine  mx = max_b m_b;  S = SUM_b s_b expf(m_b - mx);  lse = safe_log(S) + mx                        (every workgroup, same bits)
//   pass B   w_i = x_i - lse -> logw;  per block  e_b = sum expf(2 w_i);  q_i = fixed-point det_exp(w_i); in-block inclusive
//            scan of q -> cdf[i] (HBM, u64), block total, first maximum of p                              -> part[b]
//   -------- barrier --------
//   combine  nEff = 1 / (SUM_b e_b) / n, decision; block prefix of the totals (integers: associative)      (every workgroup)
//   search   a workgroup draws ONLY ITS OWN slots j: threshold T_j, block by a search over the block ends (LDS), position by a
//            search over that block's scan (8 probes, L2), copy_particles for the slot
//
// THE BITS DEPEND ON n ALONE - not on the number of workgroups, not on which of them takes a block: a block's sums are
// one fixed tree (64-lane butterfly, then the four waves in order), SUM_b is one fixed tree over the block values (lane l adds
// the blocks l, l + 64, ... in order, then the butterfly), the CDF is integers.  So the staged launch, the fused tail of the
// update kernel, every shard of a sharded filter and the gathered vector of a multi-GPU run agree bit for bit by construction,
// and the resampling indices equal the oracle's (o_resample) on the same weights as before.  (The log-sum-exp differs from the
// one-workgroup forms in its rounding - per-block maxima folded by expf(m_b - mx) - which is why the form is chosen by n alone.)
// ------------------------------------------------------------------------------------------
#define PHD_GRID_WEIGHTS_MIN 4096        // n above this takes the block form (every caller, every launch shape)
#define PHD_GW_REC 8                     // 32-bit words per block record: m, s, e2, besti, Qtot (2), best (2)

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid-wide barrier of the W workgroups of the routine: every thread has issued its sc1 stores; drain them, meet, one lane
// arrives and polls.  -> false on a time-out (~seconds: never in practice; the caller leaves without touching the counters)
__device__ __forceinline__ bool grid_weights_barrier(unsigned* ctr, unsigned W, int tid, LDS_T(int)* s_ok, unsigned* diag, unsigned code)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < W) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { ok = false; break; }
        }
        if (!ok) atomicCAS(diag, 0u, code | (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8));   // (first time-out only)
        *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return *s_ok != 0;
}

// the fixed tree over block values: lane l adds the blocks l, l + 64, ... in order, then the butterfly (one wave; every lane
// returns the result).  `term(b)` is evaluated for b < B only.
template <class F>
__device__ __forceinline__ float blocks_sum(int B, int lane, F term)
{
    float acc = 0.f;
    for (int b = lane; b < B; b += 64) acc += term(b);
    return wave_sum(acc);
}

// g: this workgroup's index among the W of the routine.  s_dyn: (B + 16) * 8 bytes of LDS.  ticket (fused step): the arrival
// counter of the particles' workgroups, reset by the last workgroup of the routine to leave.
template <bool HANDOFF>
__device__ __forceinline__ void weights_grid_body(const WeightArgs& A, int g, int W, unsigned char* s_dyn, unsigned* status, unsigned* ticket)
{
    // (PHD_T = 512 threads: two blocks of 256 per pass)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = tid >> 8, wq = wv & 3;   // wq: wave within its block
    const int n = A.n, B = (n + 255) >> 8;
    LDS_T(u64)* const s_qend = (LDS_T(u64)*)(LDS_T(unsigned char)*)s_dyn;                        // [B] inclusive block ends of the CDF
    LDS_T(float)* const s_f = (LDS_T(float)*)(s_qend + B);                                       // [16] wave partials, [16..20) globals
    LDS_T(u64)* const s_q = (LDS_T(u64)*)(s_f + 24);                                             // [8]
    LDS_T(double)* const s_bv = (LDS_T(double)*)(s_q + 8);                                       // [8]
    LDS_T(int)* const s_bi = (LDS_T(int)*)(s_bv + 8);                                            // [8] + [8] flags
    LDS_T(int)* const s_ok = s_bi + 8;
    float* const part = A.gpart;
    u64* const cdf = (u64*)A.cdf;
    const bool normalize = (A.mode & W_NORMALIZE) != 0;
    const bool may_resample = (A.mode & (W_RESAMPLE_FORCE | W_RESAMPLE_AUTO)) != 0;
    const size_t ls = A.in_stride ? (size_t)A.in_stride : 1;
    auto load_x = [&](int i) -> float {
        float x = A.logw_in[(size_t)i * ls];
        if (A.mode & W_ACCUMULATE) x += ld_f32<HANDOFF>(&A.dlogw[i]);
        return x;
    };
    // sum of a block's 256 values: butterfly per wave, then the block's four waves in order
    auto block_sum4 = [&](float v) -> float {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) s_f[wv] = v;
        __syncthreads();
        const int w0 = half * 4;
        return ((s_f[w0] + s_f[w0 + 1]) + s_f[w0 + 2]) + s_f[w0 + 3];
    };
    float lse = 0.f;
    // ---- pass A: block maxima and shifted sums (only when the vector is to be normalised)
    if (normalize) {
        for (int b0 = 2 * g; b0 < B; b0 += 2 * W) {
            const int b = b0 + half, i = 256 * b + (tid & 255);
            const bool live = b < B && i < n;
            const float x = live ? load_x(i) : -FLT_MAX;
            if (live && A.raw_out) A.raw_out[i] = x;
            float m = wave_max_f(x);
            __syncthreads();
            if (lane == 0) s_f[8 + wv] = m;
            __syncthreads();
            const int w0 = 8 + half * 4;
            m = fmaxf(fmaxf(s_f[w0], s_f[w0 + 1]), fmaxf(s_f[w0 + 2], s_f[w0 + 3]));
            const float s = block_sum4(live ? expf(x - m) : 0.f);
            if (b < B && (tid & 255) == 0) { st_sc1(&part[PHD_GW_REC * b], m); st_sc1(&part[PHD_GW_REC * b + 1], s); }
        }
        if (!grid_weights_barrier(&A.gsync[0], (unsigned)W, tid, s_ok, &A.gsync[3], 2u)) { if (tid == 0) atomicOr(status, PHD_STATUS_TAIL_TIMEOUT); return; }
        if (wv == 0) {
            float mx = -FLT_MAX;
            for (int b = lane; b < B; b += 64) mx = fmaxf(mx, ld_sc1(&part[PHD_GW_REC * b]));
            mx = wave_max_f(mx);
            const float S = blocks_sum(B, lane, [&](int b) { return ld_sc1(&part[PHD_GW_REC * b + 1]) * expf(ld_sc1(&part[PHD_GW_REC * b]) - mx); });
            if (lane == 0) s_f[16] = safe_log(S) + mx;
        }
        __syncthreads();
        lse = s_f[16];
    } else if (A.raw_out && (A.mode & W_ACCUMULATE)) {
        for (int b0 = 2 * g; b0 < B; b0 += 2 * W) {
            const int i = 256 * (b0 + half) + (tid & 255);
            if (i < n) A.raw_out[i] = load_x(i);
        }
    }
    // ---- pass B: normalised weights out, nEff terms, fixed-point CDF per block
    const int sb = cdf_scale_bits(n);
    const double scale = ldexp(1.0, sb);
    for (int b0 = 2 * g; b0 < B; b0 += 2 * W) {
        const int b = b0 + half, i = 256 * b + (tid & 255);
        const bool live = b < B && i < n;
        float w = -FLT_MAX;
        if (live) {
            w = load_x(i);
            if (normalize) w -= lse;
            if (A.logw) A.logw[i] = w;
        }
        const float e2 = block_sum4(live ? expf(2 * w) : 0.f);
        u64 q = 0, qtot = 0;
        double best = -1.0;
        int besti = 0x7FFFFFFF;
        if (may_resample) {
            if (live) {
                const double p = det_exp(w);
                q = cdf_quantise(p, scale);
                best = p; besti = i;
            }
            u64 incl = wave_incl_scan(q);
            // first maximum of p in the wave (strict '>': the lowest index among equals)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = xor_lane(best, off);
                const int oi = xor_lane(besti, off);
                if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
            }
            if (lane == 63) s_q[wv] = incl;
            if (lane == 0) { s_bv[wv] = best; s_bi[wv] = besti; }
            __syncthreads();
            const int w0 = half * 4;
            u64 before = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 c = s_q[w0 + k];
                if (k < wq) before += c;
                qtot += c;
                if (s_bv[w0 + k] > best || (s_bv[w0 + k] == best && s_bi[w0 + k] < besti)) { best = s_bv[w0 + k]; besti = s_bi[w0 + k]; }
            }
            incl += before;
            if (live) st_sc1(&cdf[i], incl);
            __syncthreads();
        }
        if (b < B && (tid & 255) == 0) {
            st_sc1(&part[PHD_GW_REC * b + 2], e2);
            st_sc1((int*)&part[PHD_GW_REC * b + 3], besti);
            st_sc1((u64*)&part[PHD_GW_REC * b + 4], qtot);
            st_sc1((u64*)&part[PHD_GW_REC * b + 6], (u64)__double_as_longlong(best));
        }
    }
    if (!grid_weights_barrier(&A.gsync[1], (unsigned)W, tid, s_ok, &A.gsync[3], 3u)) { if (tid == 0) atomicOr(status, PHD_STATUS_TAIL_TIMEOUT); return; }
    // ---- nEff and the decision (src/main.cpp:1281-1297): every workgroup, the same bits
    if (wv == 0) {
        const float s2 = blocks_sum(B, lane, [&](int b) { return ld_sc1(&part[PHD_GW_REC * b + 2]); });
        if (lane == 0) s_f[17] = s2;
    }
    // block ends of the CDF: inclusive prefix of the block totals (wave 1; integers)
    if (wv == 1 && may_resample) {
        u64 run = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            const u64 incl = wave_incl_scan(b < B ? ld_sc1((const u64*)&part[PHD_GW_REC * b + 4]) : 0ull) + run;
            if (b < B) s_qend[b] = incl;
            run = lane63_u64(incl);
        }
    }
    __syncthreads();
    const float s2 = s_f[17];
    const float neff = (float)(1.0 / (double)s2 / (double)n);
    int doit = 0;
    if (A.mode & W_RESAMPLE_FORCE) doit = 1;
    else if ((A.mode & W_RESAMPLE_AUTO) && (neff <= A.resample_thresh) && (A.mode & W_HAD_MEAS)) doit = 1;   // :1286
    if (g == 0 && tid == 0) { A.neff_out[0] = neff; A.did_resample[0] = doit; }
    const int n_new = A.n_new;
    if (!doit) {
        const int lim = (A.mode & W_COMMIT) ? n : n_new;
        for (int b0 = 2 * g; b0 < B; b0 += 2 * W) {
            const int j = 256 * (b0 + half) + (tid & 255);
            if (j < lim) {
                A.idx_out[j] = j;                                                                          // :1292-1296
                if (A.mode & W_COMMIT) { A.pose_out[j] = ld_pose<HANDOFF>(&A.pose_in[j]); A.parent_out[j] = ld_i32<HANDOFF>(&A.parent_in[j]); }
            }
        }
    } else {
        const u64 ctot = s_qend[B - 1];
        const double interval = 1.0 / n_new;
        // the overflow guard (src/main.cpp:475-494): the arg-max of p, needed only if the last threshold exceeds the total mass
        int argmax = 0;
        {
            const int jl = n_new - 1;
            const double ul = (A.n_uniforms == 1) ? A.u0 : A.uniforms[jl];
            if ((u64)ceil((jl * interval + ul * interval) * scale) > ctot) {   // uniform
                if (wv == 0) {
                    double best = -1.0;
                    int besti = 0x7FFFFFFF;
                    for (int b = lane; b < B; b += 64) {
                        const double ob = __longlong_as_double((long long)ld_sc1((const u64*)&part[PHD_GW_REC * b + 6]));
                        const int oi = ld_sc1((const int*)&part[PHD_GW_REC * b + 3]);
                        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const double ob = xor_lane(best, off);
                        const int oi = xor_lane(besti, off);
                        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
                    }
                    if (lane == 0) s_bi[0] = besti;
                }
                __syncthreads();
                argmax = s_bi[0];
            }
        }
        const float nlw = (float)(-log((double)A.n_weight_norm));
        int top = 1;
        while (top < B) top <<= 1;
        for (int b0 = 2 * g; b0 < B; b0 += 2 * W) {
            const int j = 256 * (b0 + half) + (tid & 255);
            if (j >= n_new) continue;
            const double u = (A.n_uniforms == 1) ? A.u0 : A.uniforms[j];
            const u64 T = (u64)ceil((j * interval + u * interval) * scale);                               // :468
            int idx;
            if (T > ctot) idx = argmax;                                                                    // :475-494
            else {
                // the block: blocks whose end is below T (branch-free lower bound over the block ends, LDS)
                int c = 0;
                for (int step = top; step > 0; step >>= 1) {
                    const int probe = c + step;
                    if (probe <= B && s_qend[probe - 1] < T) c = probe;
                }
                // c < B here (T <= ctot).  Inside block c: entries whose global CDF value is below T (L2, written by another
                // workgroup: sc1 loads)
                const u64 base = c ? s_qend[c - 1] : 0ull;
                const int lo = 256 * c, len = (n - lo < 256) ? (n - lo) : 256;
                int pos = 0;
#pragma unroll
                for (int step = 256; step > 0; step >>= 1) {
                    const int probe = pos + step;
                    if (probe <= len && base + ld_sc1(&cdf[lo + probe - 1]) < T) pos = probe;
                }
                idx = lo + pos;
            }
            A.idx_out[j] = idx;
            if (A.mode & W_COMMIT) {                                    // copy_particles (src/slamtypes.h:313-333)
                A.pose_out[j] = ld_pose<HANDOFF>(&A.pose_in[idx]);
                A.parent_out[j] = ld_i32<HANDOFF>(&A.parent_in[idx]);
                if (A.logw) A.logw[j] = nlw;
            }
        }
    }
    // ---- leave: the last workgroup re-arms the counters (and the fused step's ticket) for the next launch
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&A.gsync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (unsigned)W - 1u) {
            __hip_atomic_store(&A.gsync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&A.gsync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&A.gsync[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// workgroups the block form runs on: one per two blocks, at most 64 (a workgroup then loops over its blocks)
__host__ __device__ inline int grid_weights_workgroups(int n)
{
    const int w = ((n + 255) / 256 + 1) / 2;
    return w < 1 ? 1 : (w > 64 ? 64 : w);
}
__host__ __device__ inline size_t grid_weights_lds_bytes(int n) { return ((size_t)(n + 255) / 256 + 40) * 8; }

// (phd_weights_grid_kernel, the stand-alone launch of the block form: phd_kernels.hip, main translation unit)

} // namespace phd
