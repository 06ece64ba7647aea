// phd_cphd.h — the CPHD block of the update kernel (filter_type = 1).
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"
#include "phd_sort.h"
#include "phd_merge.h"
#include "phd_predict.h"

namespace phd {

// ------------------------------------------------------------------------------------------
// CPHD variant (filter_type = 1): the cardinality-dependent terms of one particle's update.
//
// The reference's HEAD has no runnable CPHD (kernels commented out, src/phdfilter.cu:701-779,
// 1360-1591); the complete statement is src/phdfilter.cu.bak — cardinalityPredictKernel :518-545,
// birth cardinality :779-790, computeEsfKernel :1191-1274, computePsiKernel :1282-1412,
// cphdUpdateKernel :1420-1462 — whose decomposition and log-domain arithmetic this follows, with
// the recursion of Vo, Vo & Cantoni (IEEE TSP 2007) stated correctly where the .bak is defective
// (see oracle/cphd_cpu.c, the CPU statement this block is tested against; parity unpinned).
//
//   predicted cardinality   prior (*) Binomial(M, birthWeight)                  thread per n
//   I_u[j]                  log sum_n p(n) P(n,j+u) Wq^(n-j-u) / W1^n            wave per j, lanes over n
//   ESF jobs                e_j(Xi) and the M leave-one-out e_j(Xi \ m): one job per wave at a time,
//                           the M-step log-domain recursion held in registers (lane <-> j),
//                           neighbours by wave shuffles; O(M^3 / 512) lse2 per thread
//   <Y0,p>, <Y1,p>, <Y1[Z\m],p>   wave reductions at the end of each job
//   updated cardinality     thread per n
// Outputs: L.logZ[m] (detection / birth weight = exp(lw - logZ[m])), the missed-detection factor
// r1 = <Y1,p>/<Y0,p>, log <Y0,p> (particle log-weight increment), cn_out[0..cn_len).
// ------------------------------------------------------------------------------------------
struct CphdLds {
    lds_f32 cnq, cnp, lfact, lxi, I0, I1, lD, efull, cnb, scal;
};
enum { CQ_LY0 = 0, CQ_LY1 = 1, CQ_R1 = 2 };

__host__ __device__ __forceinline__ u32 cphd_lds_layout(int cn_len, int MM, u32 off[10])
{
    const u32 cn = align16u(4u * (u32)cn_len);
    const u32 lf = align16u(4u * (u32)((cn_len > MM + 1 ? cn_len : MM + 1) + 1));
    const u32 mm = align16u(4u * (u32)(MM + 1));
    u32 p = 0;
    off[0] = p; p += cn;  // cnq
    off[1] = p; p += cn;  // cnp
    off[2] = p; p += lf;  // lfact
    off[3] = p; p += mm;  // lxi
    off[4] = p; p += mm;  // I0
    off[5] = p; p += mm;  // I1
    off[6] = p; p += mm;  // lD
    off[7] = p; p += mm;  // efull
    off[8] = p; p += mm;  // cnb
    off[9] = p; p += 64u; // scal
    return p;
}


__device__ __forceinline__ CphdLds cphd_carve(lds_u8 base, int cn_len, int MM)
{
    u32 off[10];
    cphd_lds_layout(cn_len, MM, off);
    CphdLds Q;
    Q.cnq = (lds_f32)(base + off[0]); Q.cnp = (lds_f32)(base + off[1]); Q.lfact = (lds_f32)(base + off[2]);
    Q.lxi = (lds_f32)(base + off[3]); Q.I0 = (lds_f32)(base + off[4]); Q.I1 = (lds_f32)(base + off[5]);
    Q.lD = (lds_f32)(base + off[6]); Q.efull = (lds_f32)(base + off[7]); Q.cnb = (lds_f32)(base + off[8]);
    Q.scal = (lds_f32)(base + off[9]);
    return Q;
}


__device__ __forceinline__ float lse2f(float a, float b)
{
    const float mx = a > b ? a : b, mn = a > b ? b : a;
    return mx + log1pf(expf(mn - mx));
}

__device__ __forceinline__ float clamp_log(float x) { return x < -1e30f ? -1e30f : x; }

// the sweeps of cphd_block are specialised on the number of 64-lane tiles (M <= 64 TILES): with TILES = 1 — every
// configuration of BASELINE.json — the tile loops and their guards fold away, which halves the instruction count
// of these latency-bound sections
// PHD_FW waves run the (cheap) backward recursion redundantly and each takes the inner products of every PHD_FW-th step:
// PHD_FW inner products (two wave reductions and a log each) in flight, no data exchanged between the waves (eight waves
// are no faster than four: the redundant recursion is issue capacity the other resident workgroup can use)
#ifndef PHD_FW
#define PHD_FW 4
#endif
// ------------------------------------------------------------------------------------------
// The two sweeps in the order that lets the first one run BESIDE the cardinality work: the forward recursion
// P_{m+1} = P_m (1 + xi_m x) needs only the roots, so one wave runs it — parking the rows P_m[0..m] in the HBM scratch —
// while the other seven compute the predicted cardinality and the n-sums; the backward recursion T_m = T_{m+1} + xi_m
// shift(T_{m+1}), T_M = c, needs the n-sums (c_j = exp(I1[j]) lambda^(M-1-j) e^-lambda) and takes the inner products
// D_m = <P_m, T_{m+1}> against the parked rows on the way (PHD_FW waves run the cheap recursion redundantly, each takes
// every PHD_FW-th inner product).  (Round 1 parked T and carried P: the same numbers, but both sweeps then had to wait
// for the n-sums.)
// ------------------------------------------------------------------------------------------
template <int tiles>
__device__ __forceinline__ void cphd_esf_forward_park(const CphdLds& Q, float2* __restrict__ P_scratch, int M, int lane)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    const int XF_ZERO_K = -(1 << 28);
    // P_m[a], a = lane + 1 + 64 c in registers (P_m[0] = 1 is implicit)
    float pm[4] = {0.f, 0.f, 0.f, 0.f};
    int pk[4] = {XF_ZERO_K, XF_ZERO_K, XF_ZERO_K, XF_ZERO_K};
    const float xv = (tiles == 1 && lane < M) ? Q.lxi[lane] : 0.f;
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);   // a chain of M dependent steps on one wave: let it issue ahead of the SIMD's other waves
#endif
    for (int m = 0; m < M; ++m) {
        // park row m: P_m[0..m] ([0] = 1 = 0.5 * 2^1)
        float2* row = P_scratch + (size_t)m * M;
        if (lane == 0) row[0] = make_float2(0.5f, __int_as_float(1));
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < tiles && lane + 1 + 64 * c <= m) row[lane + 1 + 64 * c] = make_float2(pm[c], __int_as_float(pk[c]));
        const float x = (tiles == 1) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), m)) : Q.lxi[m];
        // P_{m+1}[a] = P_m[a] + xi_m P_m[a-1]
        float um[4];
        int uk[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            um[c] = 0.f; uk[c] = XF_ZERO_K;
            if (c < tiles) {
                const float up_m = lane_up1(pm[c]);
                const int up_k = lane_up1(pk[c]);
                const float cm = (c > 0) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm[c > 0 ? c - 1 : 0]), 63))
                                         : 0.5f;                                       // P[0] = 1 = 0.5 * 2^1
                const int ck = (c > 0) ? __builtin_amdgcn_readlane(pk[c > 0 ? c - 1 : 0], 63) : 1;
                um[c] = (lane == 0) ? cm : up_m;
                uk[c] = (lane == 0) ? ck : up_k;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < tiles && lane + 64 * c <= m) {
                const float pr = um[c] * x;
                const int k = pk[c] > uk[c] ? pk[c] : uk[c];
                const float s2 = ldexpf(pm[c], pk[c] - k) + ldexpf(pr, uk[c] - k);
                int dk = 0;
                pm[c] = frexpf(s2, &dk);
                pk[c] = k + dk;
            }
    }
    // full set: log e_j = log P_M[j]
    if (lane == 0) Q.efull[0] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < tiles && lane + 1 + 64 * c <= M)
            Q.efull[lane + 1 + 64 * c] = pm[c] > 0.f ? logf(pm[c]) + (float)pk[c] * 0.69314718f : LOG0F;
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __threadfence(); // the rows are read back by the other waves of this workgroup (after its barrier)
}

// <Y0,p> and <Y1,p> from the full-set ESFs (one wave)
__device__ __forceinline__ void cphd_full_set(const CphdLds& Q, int M, int lane, float llam, float lam)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    float mx0 = LOG0F, mx1 = LOG0F;
    for (int j = lane; j <= M; j += 64) {
        const float kterm = (float)(M - j) * llam - lam;   // (M-j)! p_K(M-j), Poisson clutter (.bak:398-400)
        mx0 = fmaxf(mx0, Q.efull[j] + Q.I0[j] + kterm);
        mx1 = fmaxf(mx1, Q.efull[j] + Q.I1[j] + kterm);
    }
    mx0 = wave_max_f(mx0); mx1 = wave_max_f(mx1);
    float s0 = 0.f, s1 = 0.f;
    for (int j = lane; j <= M; j += 64) {
        const float kterm = (float)(M - j) * llam - lam;
        s0 += expf(Q.efull[j] + Q.I0[j] + kterm - mx0);
        s1 += expf(Q.efull[j] + Q.I1[j] + kterm - mx1);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { Q.scal[CQ_LY0] = safe_log(s0) + mx0; Q.scal[CQ_LY1] = safe_log(s1) + mx1; }
}

template <int tiles>
__device__ __forceinline__ void cphd_esf_backward_dot(const CphdLds& Q, const float2* __restrict__ P_scratch, int M, int lane,
                                                      int wave, float llam, float lam)
{
#pragma clang fp contract(off)
    if (wave >= PHD_FW) return;   // the recursion is redundant work: only this many waves take part
    const float LOG0F = -FLT_MAX;
    const int XF_ZERO_K = -(1 << 28);
    float tm[4];
    int tk[4];
    // T_M[a] = c_a, a = lane + 64 c
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int a = lane + 64 * c;
        tm[c] = 0.f; tk[c] = XF_ZERO_K;
        if (c < tiles && a < M) {
            const float Lg = Q.I1[a] + ((float)(M - 1 - a) * llam - lam);
            if (Lg > -1e30f) {
                const double t = (double)Lg * 1.4426950408889634;
                const double kf = ceil(t);
                tm[c] = (float)exp2(t - kf);
                tk[c] = (int)kf;
            }
        }
    }
    // this wave's steps: m = M - 1 - wave - PHD_FW u, u = 0, 1, ...; PF of their rows in flight
    constexpr int PF = 4;
    float2 rbuf[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int mu = M - 1 - wave - PHD_FW * u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rbuf[u][c] = make_float2(0.f, 0.f);
            if (mu >= 0 && c < tiles && lane + 64 * c <= mu) rbuf[u][c] = P_scratch[(size_t)mu * M + lane + 64 * c];
        }
    }
    const float xv = (tiles == 1 && lane < M) ? Q.lxi[lane] : 0.f;
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int mb = M - 1; mb >= 0; mb -= PF * PHD_FW) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
    for (int r = 0; r < PHD_FW; ++r) {
        const int m = mb - PHD_FW * u - r;
        if (m >= 0) {
        if (r == wave) {
            // D_m = sum_{a=0..m} P_m[a] T_{m+1}[a]
            float qm[4];
            int qk[4];
            int kmax = 2 * XF_ZERO_K;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int a = lane + 64 * c;
                qm[c] = 0.f; qk[c] = 2 * XF_ZERO_K;
                if (c < tiles && a <= m) {
                    const float2 pr = rbuf[u][c];
                    qm[c] = pr.x * tm[c];
                    qk[c] = __float_as_int(pr.y) + tk[c];
                }
                kmax = max(kmax, qk[c]);
            }
            const int mn = m - PF * PHD_FW;   // refill this slot with the row of this wave's step PF turns ahead
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (mn >= 0 && c < tiles && lane + 64 * c <= mn) rbuf[u][c] = P_scratch[(size_t)mn * M + lane + 64 * c];
            kmax = wave_max_i(kmax);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) s += ldexpf(qm[c], qk[c] - kmax);
            s = wave_sum(s);
            if (lane == 0) {
                int dk = 0;
                const float dm = frexpf(s, &dk);
                Q.lD[m] = dm > 0.f ? logf(dm) + (float)(kmax + dk) * 0.69314718f : LOG0F;   // log <Y1[Z \ m], p>
            }
        }
        if (m >= 1) {
            // T_m[a] = T_{m+1}[a] + xi_m T_{m+1}[a+1], a <= m - 1
            const float x = (tiles == 1) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), m)) : Q.lxi[m];
            float nm[4];
            int nk[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                nm[c] = 0.f; nk[c] = XF_ZERO_K;
                if (c < tiles) {
                    const float dn_m = lane_down1(tm[c]);
                    const int dn_k = lane_down1(tk[c]);
                    const float cm = (c < 3) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tm[c < 3 ? c + 1 : 3]), 0)) : 0.f;
                    const int ck = (c < 3) ? __builtin_amdgcn_readlane(tk[c < 3 ? c + 1 : 3], 0) : XF_ZERO_K;
                    nm[c] = (lane == 63) ? cm : dn_m;
                    nk[c] = (lane == 63) ? ck : dn_k;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int a = lane + 64 * c;
                if (c < tiles && a <= m - 1) {
                    const float pr = nm[c] * x;
                    const int k = tk[c] > nk[c] ? tk[c] : nk[c];
                    const float s = ldexpf(tm[c], tk[c] - k) + ldexpf(pr, nk[c] - k);
                    int dk = 0;
                    tm[c] = frexpf(s, &dk);
                    tk[c] = k + dk;
                }
            }
        }
        } // m >= 0
    }
    }
    }
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// barrier among a subset of the workgroup's waves (the hardware barrier counts all of them): an LDS arrival counter,
// polled.  Every participating wave calls it the same number of times; `target` is its running arrival count.
__device__ __forceinline__ void waves_sync(LDS_T(int)* ctr, int n_waves, int& target, int lane)
{
    target += n_waves;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add((int*)ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load((int*)ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the same sums with the n-dependent part taken out of the loop: with B_n = log p(n) + log n! + n (log Wq - log W1)
// (kept in the cnq array, free once the predicted cardinality exists), the term is B_n - log (n-j)! - j log Wq: two LDS
// reads and one subtraction per term instead of three reads and six operations.  Needs finite log Wq, log W1 (an empty
// map has Wq = 0: the caller then takes cphd_nsums, whose 0 * (-1e30) products are exact).
template <int CH>
__device__ __forceinline__ void cphd_nsums_fast(const CphdLds& Q, int M, int Nmax, int lane, int wave, int n_waves, float lWq)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    for (int j = wave; j <= M + 1; j += n_waves) {
        float tv[CH];
        float mx = LOG0F;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int n = j + lane + 64 * c;
            tv[c] = LOG0F;
            if (n <= Nmax) {
                tv[c] = Q.cnq[n] - Q.lfact[n - j];
                mx = fmaxf(mx, tv[c]);
            }
        }
        mx = wave_max_f(mx);
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (j + lane + 64 * c <= Nmax) sacc += __expf(tv[c] - mx);
        sacc = wave_sum(sacc);
        if (lane == 0) {
            const float v = (j <= Nmax) ? (safe_log(sacc) + mx) - (float)j * lWq : LOG0F;
            if (j <= M) Q.I0[j] = v;
            if (j >= 1) Q.I1[j - 1] = v;
        }
    }
}

template <int CH>
__device__ __forceinline__ void cphd_nsums(const CphdLds& Q, int M, int Nmax, int lane, int wave, int n_waves, float lWq, float lW1)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    for (int j = wave; j <= M + 1; j += n_waves) {
        float tv[CH];
        float mx = LOG0F;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int n = j + lane + 64 * c;
            tv[c] = LOG0F;
            if (n <= Nmax) {
                tv[c] = Q.cnp[n] + (Q.lfact[n] - Q.lfact[n - j]) + (float)(n - j) * lWq - (float)n * lW1;
                mx = fmaxf(mx, tv[c]);
            }
        }
        mx = wave_max_f(mx);
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (j + lane + 64 * c <= Nmax) sacc += __expf(tv[c] - mx);
        sacc = wave_sum(sacc);
        if (lane == 0) {
            const float v = (j <= Nmax) ? safe_log(sacc) + mx : LOG0F;
            if (j <= M) Q.I0[j] = v;
            if (j >= 1) Q.I1[j - 1] = v;
        }
    }
}

__device__ __forceinline__ void cphd_block(const Lds& L, const CphdLds& Q, const DevConfig& cfg, int M, int MM, int cn_len,
                                        const float* __restrict__ lfact_g, int lfact_len, const float* __restrict__ cn_prior,
                                        float* __restrict__ cn_out, float2* __restrict__ T_scratch, float w_all, float pdw,
                                        int tid, u64* cq)
{
#pragma clang fp contract(off)
    // cq (diagnostic instantiation, thread 0): time of [staging + birth cardinality, forward sweep beside the cardinality work,
    // backward sweep + inner products, rest]
#define CQSTAMP(k) do { if (cq && tid == 0) cq[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
    CQSTAMP(0);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const int Nmax = cn_len - 1;
    const float lam = cfg.clutterRate;
    const float llam = safe_log(lam), lkap = safe_log(cfg.clutterDensity);
    const float lbw = safe_log(cfg.birthWeight), l1bw = safe_log(1 - cfg.birthWeight);
    const float W1 = w_all + (float)M * cfg.birthWeight;   // <1,v>: map + births
    const float Wq = w_all - pdw;                          // <1-pD,v>: a birth is always detected
    const float lW1 = clamp_log(safe_log(W1)), lWq = clamp_log(safe_log(Wq));
    const float LOG0F = -FLT_MAX;

    for (int i = tid; i < cn_len; i += PHD_T) Q.cnq[i] = cn_prior[i];
    for (int i = tid; i < lfact_len; i += PHD_T) Q.lfact[i] = lfact_g[i];
    __syncthreads();
    // birth cardinality: Binomial(k; M, birthWeight) (.bak:779-790)
    const int Kb = M < Nmax ? M : Nmax;
    for (int k = tid; k <= Kb; k += PHD_T)
        Q.cnb[k] = Q.lfact[M] - Q.lfact[k] - Q.lfact[M - k] + (float)k * lbw + (float)(M - k) * l1bw;
    __syncthreads();
    // From here wave 0 runs the forward ESF sweep (it needs only the roots) while waves 1..7 do the cardinality work —
    // predicted cardinality, B_n, n-sums — synchronising among themselves through an LDS arrival counter.
    const int tiles = (M + 63) >> 6;
    const bool finite_w = lWq > -1e29f && lW1 > -1e29f;     // (an empty map has Wq = 0, an empty map without births W1 = 0)
    CQSTAMP(1);
    if (wave == 0) {
        if (tiles == 1) cphd_esf_forward_park<1>(Q, T_scratch, M, lane);
        else if (tiles == 2) cphd_esf_forward_park<2>(Q, T_scratch, M, lane);
        else cphd_esf_forward_park<4>(Q, T_scratch, M, lane);
    } else {
        LDS_T(int)* sctr = (LDS_T(int)*)&L.ctr[CTR_WSYNC];
        int target = 0;
        const int t7 = tid - 64, T7 = PHD_T - 64, w7 = wave - 1, W7 = PHD_NW - 1;
        // predicted cardinality (.bak:518-545)
        for (int n = t7; n <= Nmax; n += T7) {
            const int kmax = n < Kb ? n : Kb;
            float mx = Q.cnb[0] + Q.cnq[n];
            for (int k = 1; k <= kmax; ++k) mx = fmaxf(mx, Q.cnb[k] + Q.cnq[n - k]);
            float s = 0.f;
            for (int k = 0; k <= kmax; ++k) s += __expf(Q.cnb[k] + Q.cnq[n - k] - mx);
            Q.cnp[n] = safe_log(s) + mx;
        }
        waves_sync(sctr, W7, target, lane);
        // I_u[j] = log sum_n p(n) P(n,j+u) Wq^(n-j-u) / W1^n.  Since P(n,j+1) Wq^(n-j-1) is the u = 0 term of j+1,
        // I_1[j] = I_0[j+1] (the same floating-point expression): one family J[j] = I_0[j], j = 0..M+1.
        // Wave per j, lanes over n; the terms stay in registers between the max and the sum pass (n <= 1023).
        // (the chunk count is a compile-time constant per cardinality length: max_cardinality 255 needs 4 of the 16)
        if (finite_w) {
            for (int n = t7; n <= Nmax; n += T7) Q.cnq[n] = Q.cnp[n] + Q.lfact[n] + (float)n * (lWq - lW1);   // B_n
            waves_sync(sctr, W7, target, lane);
            if (cn_len <= 256) cphd_nsums_fast<4>(Q, M, Nmax, lane, w7, W7, lWq);
            else if (cn_len <= 512) cphd_nsums_fast<8>(Q, M, Nmax, lane, w7, W7, lWq);
            else cphd_nsums_fast<16>(Q, M, Nmax, lane, w7, W7, lWq);
        } else {
            if (cn_len <= 256) cphd_nsums<4>(Q, M, Nmax, lane, w7, W7, lWq, lW1);
            else if (cn_len <= 512) cphd_nsums<8>(Q, M, Nmax, lane, w7, W7, lWq, lW1);
            else cphd_nsums<16>(Q, M, Nmax, lane, w7, W7, lWq, lW1);
        }
    }
    __syncthreads();
    CQSTAMP(2);
    // ESFs (.bak:1224-1272).  The .bak runs one full recursion per left-out measurement (O(M^3)); here
    //   e(Xi \ m) = P_m (*) S_{m+1}   (ESFs of the roots before and after m), so
    //   <Y1[Z\m],p> = sum_a P_m[a] T_{m+1}[a],  T_{m+1}[a] = sum_b S_{m+1}[b] c_{a+b},  c_j = exp(I1[j]) lambda^(M-1-j) e^-lambda
    // and T obeys the same one-root recursion run backwards, T_m[a] = T_{m+1}[a] + xi_m T_{m+1}[a+1], T_M = c:
    // O(M^2), all terms positive.  The rows P_m[0..m] were parked in HBM scratch by the forward sweep above (M^2 x 8 B
    // per particle — what 288 GB are for; they come back out of L2); the backward sweep carries T in registers and takes
    // one inner product per measurement.  Values span hundreds of decades, so each is a float mantissa with its own
    // integer exponent (m 2^k): align with v_ldexp, renormalise with v_frexp — exact operations around one correctly
    // rounded multiply and add (the oracle does the same).  Wave PHD_FW, idle in the sweep, takes <Y0,p> and <Y1,p>.
    if (wave == PHD_FW) cphd_full_set(Q, M, lane, llam, lam);
    if (tiles == 1) cphd_esf_backward_dot<1>(Q, T_scratch, M, lane, wave, llam, lam);
    else if (tiles == 2) cphd_esf_backward_dot<2>(Q, T_scratch, M, lane, wave, llam, lam);
    else cphd_esf_backward_dot<4>(Q, T_scratch, M, lane, wave, llam, lam);
    __syncthreads();
    CQSTAMP(3);
    const float lY0 = Q.scal[CQ_LY0];
    for (int m = tid; m < M; m += PHD_T) L.logZ[m] = -((llam - lkap) + Q.lD[m] - lY0);      // .bak:1434-1437
    if (tid == 0) Q.scal[CQ_R1] = expf(Q.scal[CQ_LY1] - lY0);                               // .bak:1452-1455
    // updated cardinality (.bak:1409-1411): p(n) Y0(n) / <Y0,p>.  With B_n as above and a_j = log e_j + (M-j) log lambda
    // - lambda - j log Wq (in the cnb array, free by now) the term is B_n + a_j - log (n-j)!: the sum over j costs two LDS
    // reads and a subtraction per term
    if (finite_w) {
        for (int j = tid; j <= M; j += PHD_T) Q.cnb[j] = Q.efull[j] + ((float)(M - j) * llam - lam) - (float)j * lWq;
        __syncthreads();
        for (int n = tid; n <= Nmax; n += PHD_T) {
            const int jmax = n < M ? n : M;
            float mx = LOG0F;
            for (int j = 0; j <= jmax; ++j) mx = fmaxf(mx, Q.cnb[j] - Q.lfact[n - j]);
            float s = 0.f;
            for (int j = 0; j <= jmax; ++j) s += __expf((Q.cnb[j] - Q.lfact[n - j]) - mx);
            cn_out[n] = Q.cnq[n] + (safe_log(s) + mx) - lY0;
        }
    } else
    for (int n = tid; n <= Nmax; n += PHD_T) {
        const int jmax = n < M ? n : M;
        float mx = LOG0F;
        for (int j = 0; j <= jmax; ++j) {
            const float t = Q.efull[j] + ((float)(M - j) * llam - lam) + (Q.lfact[n] - Q.lfact[n - j]) + (float)(n - j) * lWq
                            - (float)n * lW1;
            mx = fmaxf(mx, t);
        }
        float s = 0.f;
        for (int j = 0; j <= jmax; ++j) {
            const float t = Q.efull[j] + ((float)(M - j) * llam - lam) + (Q.lfact[n] - Q.lfact[n - j]) + (float)(n - j) * lWq
                            - (float)n * lW1;
            s += __expf(t - mx);
        }
        cn_out[n] = Q.cnp[n] + (safe_log(s) + mx) - lY0;
    }
    __syncthreads();
    CQSTAMP(4);
#undef CQSTAMP
}

} // namespace phd
