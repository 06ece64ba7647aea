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
    lds_i32 kp;       // f64 sweeps: the block exponent of the parked row P_m
    lds_f32 zscr;     // 32 MM bytes: what the block calls L.zpart (the kernel points L.zpart here: pass 1's partial sums, then the
                      // block's rows of doubles — the common layout keeps only max(MM, 32) floats for the sums, phd_lds.h)
};
enum { CQ_LY0 = 0, CQ_LY1 = 1, CQ_R1 = 2 };

// The block's arrays: `scal` outlives the block (the missed-detection factor and log <Y0,p> are read at the emission, the
// hand-off and the tail) and sits behind the common layout; everything else is dead when the block returns.  off[9] = scal.
__host__ __device__ __forceinline__ u32 cphd_lds_layout(int cn_len, int MM, u32 off[12])
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
    off[10] = p; p += mm; // kp
    off[11] = p; p += align16u(32u * (u32)MM); // zscr
    off[9] = 0;           // scal: its own 64 bytes (cphd_carve)
    return p;
}
#define PHD_CPHD_PERSIST_BYTES 64u

// The f64 sweeps (M <= 64) park the rows P_m[0..m], m < M, in the survivor planes — empty until this block has produced the
// weights.  TRIANGULAR: row m starts at entry m (m + 1) / 2; the sweep still stores all 64 lanes of a row (zeros beyond
// entry m), which run into the places of the rows behind it — written later, in ascending m.  (M - 1) M / 2 + 64 entries:
// 16.3 KB at M = 64 instead of the 32 KB of the rectangular form, which leaves the planes room for the block's own arrays:
// at 4096 x 256 x 64 the CPHD filter then needs the common layout + 64 B, and three workgroups share a CU.
__host__ __device__ __forceinline__ u32 cphd_rows_bytes(int MM)
{
    const int m = MM < 64 ? MM : 64;
    return 8u * (u32)((m - 1) * m / 2 + 64);
}
__host__ __device__ __forceinline__ bool cphd_block_in_planes(int S_cap, int cn_len, int MM)
{
    u32 off[12];
    return 32u * (u32)S_cap >= cphd_rows_bytes(MM) + cphd_lds_layout(cn_len, MM, off);
}
// bytes behind the common layout
__host__ __device__ __forceinline__ u32 cphd_extra_lds_bytes(int S_cap, int cn_len, int MM)
{
    u32 off[12];
    const u32 blk = cphd_lds_layout(cn_len, MM, off);
    return PHD_CPHD_PERSIST_BYTES + (cphd_block_in_planes(S_cap, cn_len, MM) ? 0u : blk);
}

// persist: behind the common layout; planes: the survivor planes' base
__device__ __forceinline__ CphdLds cphd_carve(lds_u8 persist, lds_u8 planes, int S_cap, int cn_len, int MM)
{
    u32 off[12];
    cphd_lds_layout(cn_len, MM, off);
    lds_u8 base = cphd_block_in_planes(S_cap, cn_len, MM) ? planes + cphd_rows_bytes(MM) : persist + PHD_CPHD_PERSIST_BYTES;
    CphdLds Q;
    Q.cnq = (lds_f32)(base + off[0]); Q.cnp = (lds_f32)(base + off[1]); Q.lfact = (lds_f32)(base + off[2]);
    Q.lxi = (lds_f32)(base + off[3]); Q.I0 = (lds_f32)(base + off[4]); Q.I1 = (lds_f32)(base + off[5]);
    Q.lD = (lds_f32)(base + off[6]); Q.efull = (lds_f32)(base + off[7]); Q.cnb = (lds_f32)(base + off[8]);
    Q.scal = (lds_f32)persist;
    Q.kp = (lds_i32)(base + off[10]);
    Q.zscr = (lds_f32)(base + off[11]);
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

// ------------------------------------------------------------------------------------------
// Round 3: the two sweeps in DOUBLE for scans of up to 64 measurements (every BASELINE.json configuration).
// The (float mantissa, int exponent) pairs above cost ~30 instructions per recursion step — align, add, renormalise for every
// lane and every step — and each sweep is a chain of M dependent steps.  A double carries 11 bits of exponent itself: the
// step is ONE fma per lane, P_{m+1}[a] = fma(xi_m, P_m[a-1], P_m[a]), and the range beyond 2^+-1023 is kept by a block
// exponent per row, renormalised every sixteenth step (largest exponent of the row -> 2^0; entries 2^-1074 below the row's
// largest flush to zero, far below anything a sum of 64 terms can see).  53-bit mantissas instead of 24: the oracle
// (oracle/cphd_cpu.c) takes the same recursion in double and is no longer bit-identical to the device in the ESFs — the
// two agree to ~1e-15 before the logarithm rounds to float.  The rows P_m are parked in LDS when the survivor arrays (idle
// during this block) can hold them (M * 64 doubles <= 32 S bytes), else in the HBM scratch.
// ------------------------------------------------------------------------------------------
#define PHD_F64_RENORM 16      // steps between rescalings: a step grows a row by at most (1 + xi) < 2^40, 16 of them stay below 2^1023
#define PHD_FW64 4             // waves that run the (cheap) backward recursion redundantly (one per SIMD), each taking every PHD_FW64-th inner product

__device__ __forceinline__ double lane_up1(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = lane_up1((u32)b), hi = lane_up1((u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ double lane_down1(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = lane_down1((u32)b), hi = lane_down1((u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#define PHD_XD(k) { const u64 b = (u64)__double_as_longlong(v); const u32 lo = xor_lane_c<k>((u32)b), hi = xor_lane_c<k>((u32)(b >> 32)); \
                    v += __longlong_as_double((long long)(((u64)hi << 32) | lo)); }
    PHD_XD(32) PHD_XD(16) PHD_XD(8) PHD_XD(4) PHD_XD(2) PHD_XD(1)
#undef PHD_XD
    return v;
}
// eight per-lane doubles summed over the wave in one transposing pass (the double twin of reduce8_over_wave, phd_pass1.h:
// every level halves the values a lane carries — one swap per 32-bit half and one add serve two values): lane l ends with the
// wave total of a[l >> 3].  ~40 instructions for eight sums instead of 8 x 18.
__device__ __forceinline__ double swap_add32(double x, double y)
{
    // lanes < 32 get x_own + x_partner... (v_permlane32_swap exchanges the upper half of the first operand with the lower half of the second)
    const u64 bx = (u64)__double_as_longlong(x), by = (u64)__double_as_longlong(y);
    const u32x2_t lo = __builtin_amdgcn_permlane32_swap((u32)bx, (u32)by, false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane32_swap((u32)(bx >> 32), (u32)(by >> 32), false, false);
    return __longlong_as_double((long long)(((u64)hi.x << 32) | lo.x)) + __longlong_as_double((long long)(((u64)hi.y << 32) | lo.y));
}
__device__ __forceinline__ double swap_add16(double x, double y)
{
    const u64 bx = (u64)__double_as_longlong(x), by = (u64)__double_as_longlong(y);
    const u32x2_t lo = __builtin_amdgcn_permlane16_swap((u32)bx, (u32)by, false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane16_swap((u32)(bx >> 32), (u32)(by >> 32), false, false);
    return __longlong_as_double((long long)(((u64)hi.x << 32) | lo.x)) + __longlong_as_double((long long)(((u64)hi.y << 32) | lo.y));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = dpp_mov<CTRL, 0xF>((u32)b, (u32)b), hi = dpp_mov<CTRL, 0xF>((u32)(b >> 32), (u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ double reduce8_over_wave_d(double (&a)[8], int lane)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = swap_add32(a[i], a[i + 4]);      // lanes < 32 go on with values 0..3, lanes >= 32 with 4..7
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = swap_add16(a[i], a[i + 2]);      // even rows of 16 lanes: value i, odd rows: value i + 2
    const bool hi8 = (lane & 8) != 0;
    const double keep = hi8 ? a[1] : a[0], send = hi8 ? a[0] : a[1];
    double v = keep + dpp_d<0x128>(send);                                // row_ror:8
    v += dpp_d<0x141>(v);                                                // row_half_mirror: l <-> 7 - l
    v += dpp_d<0x1B>(v);                                                 // quad_perm [3,2,1,0]
    v += dpp_d<0xB1>(v);                                                 // quad_perm [1,0,3,2]
    return v;
}

// biased exponent of a non-negative double (0 for zero / denormals)
__device__ __forceinline__ int dexp_field(double v) { return (int)(((u64)__double_as_longlong(v) >> 52) & 0x7FFull); }
// log of m 2^k for a positive double m, as a float (the form the float sweeps use: log of the mantissa + k ln 2)
__device__ __forceinline__ float log_scaled(double v, int k)
{
    if (!(v > 0.0)) return -FLT_MAX;
    int e = 0;
    const double mant = frexp(v, &e);
    return logf((float)mant) + (float)(e + k) * 0.69314718f;
}

// value of lane - 1 / lane + 1 with ZERO for the lane that has no such neighbour (DPP wave shifts, bound_ctrl)
__device__ __forceinline__ double lane_up1_z(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = dpp_zero<0x138, 0xF>((u32)b), hi = dpp_zero<0x138, 0xF>((u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ double lane_down1_z(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = dpp_zero<0x130, 0xF>((u32)b), hi = dpp_zero<0x130, 0xF>((u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

// wave 0: forward sweep, rows P_m[0..m] parked as doubles (row m at rows + rs m) with their block exponent in Q.kp[m].
// The sweep wave shares its SIMD with three others and gets a quarter of the issue slots whatever its priority, so the
// time of the chain IS its instruction count: lane <-> a (P_m[a], a = 0..63; P_m[0] = 2^-kP sits in lane 0 and stays there,
// because its shifted-in neighbour is zero), no masks (entries beyond a = m are exact zeros: 0 + xi * 0), unconditional
// stores — one readlane, one convert, two DPP moves, one fma and one store per step.  P_M[64] (only M = 64 needs it, for
// the full-set ESF) is xi_63 * P_63[63].  RowPtr: an LDS or a global pointer (a generic one would compile to flat stores).
// TRI: the triangular row layout of the LDS rows (cphd_rows_bytes; h = 0 there)
template <bool TRI> __device__ __forceinline__ size_t cphd_row_off(int m, int rs) { return TRI ? (size_t)(m * (m + 1) / 2) : (size_t)m * rs; }
template <bool TRI, typename RowPtr>
__device__ __forceinline__ void cphd_esf_forward_park_f64(const CphdLds& Q, RowPtr rows, int rs, int M, int h, int lane)
{
#pragma clang fp contract(off)
    double P = lane == 0 ? 1.0 : 0.0;     // P_0 = [1]
    double top = 0.0;                     // P_M[64]
    int kP = 0;
    const float xv = lane < M ? Q.lxi[lane] : 0.f;
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int m0 = 0; m0 < M; m0 += PHD_F64_RENORM) {
        // (only the rows m >= h are parked — the inner products of the measurements below h come from the suffix rows)
        if (lane < PHD_F64_RENORM && m0 + lane < M && m0 + lane >= h) Q.kp[m0 + lane] = kP;   // the block exponent of the next rows
        const int m1 = (m0 + PHD_F64_RENORM < M) ? m0 + PHD_F64_RENORM : M;
        for (int m = m0; m < m1; ++m) {
            if (m >= h && lane < rs) rows[cphd_row_off<TRI>(m - h, rs) + lane] = P;  // row m: P_m[0..m] (zeros beyond; rs >= M > m)
            const double x = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), m));
            if (m == 63) top = x * P;                                             // (lane 63 holds P_63[63])
            P = __builtin_fma(x, lane_up1_z(P), P);                               // P_{m+1}[a] = P_m[a] + xi_m P_m[a-1]
        }
        const int r = wave_max_i(dexp_field(P)) - 1023;                          // rescale: the row's largest exponent -> 0
        P = __builtin_ldexp(P, -r);
        top = __builtin_ldexp(top, -r);
        kP += r;
    }
    // full set: log e_j = log P_M[j]
    if (lane == 0) Q.efull[0] = 0.f;                                              // e_0 = 1
    else if (lane <= M) Q.efull[lane] = log_scaled(P, kP);
    if (M == 64 && lane == 63) Q.efull[64] = log_scaled(top, kP);
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __threadfence();
}

// wave 0, after its forward sweep: the SUFFIX polynomial S_h = prod_{i >= h} (1 + xi_i x) (degree M - h) by the same step with
// the roots M-1 .. h in reverse order — M - h steps — left in an LDS row with its block exponent.  It lets the second half of
// the backward recursion start from T_h = S_h (correlated with) c without waiting for the first half.
__device__ __forceinline__ void cphd_esf_suffix_f64(const CphdLds& Q, LDS_T(double)* sh_row, LDS_T(int)* sh_exp, int M, int h, int lane,
                                                    LDS_T(int)* sctr, int conv_done)
{
#pragma clang fp contract(off)
    double S = lane == 0 ? 1.0 : 0.0;
    int kS = 0, since = 0;
    const float xv = lane < M ? Q.lxi[lane] : 0.f;
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int j = M - 1; j >= h; --j) {
        const double x = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), j));
        S = __builtin_fma(x, lane_up1_z(S), S);                                     // S_j = S_{j+1} (1 + xi_j x)
        if (++since == PHD_F64_RENORM) {
            const int r = wave_max_i(dexp_field(S)) - 1023;
            S = __builtin_ldexp(S, -r);
            kS += r;
            since = 0;
        }
    }
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // the row goes into the zpart array, which the cardinality waves read (the scaled prior) until their convolution is done:
    // wait for their arrival counter (long since reached: the convolution takes 4 us, the two sweeps 9)
    while (__hip_atomic_load((int*)sctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < conv_done) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < M) sh_row[lane] = S;                                                 // S_h[0 .. M - h] (zeros beyond)
    if (lane == 0) sh_exp[0] = kS;
}

// The inner products D_m = <Y1[Z \ m], p> = <P_m, T_{m+1}>, T_{m+1}[a] = sum_b S_{m+1}[b] c_{a+b}, from TWO chains of half the
// length that run side by side (waves 0-1 and 2-3, one wave per SIMD; within a pair the waves run the cheap recursion
// redundantly and take alternate inner products, the row of the next one requested a turn ahead):
//   upper (m >= h): T_M = c, T_m[a] = T_{m+1}[a] + xi_m T_{m+1}[a+1] backwards from M - 1 to h;
//   lower (m <  h): the same recursion from T_h, which does not wait for the upper chain: T_h[a] = sum_b S_h[b] c[a+b] is an
//                   (M - h + 1)-term correlation of c with the suffix polynomial the forward wave left behind — independent
//                   fmas, not a chain.
// (The step shifts towards lane 0 with a zero shifted in: c_k = 0 beyond k = M - 1, so every lane stays a valid entry.)
// Round 3a ran ONE chain of M steps: 9.5 us.  The logarithms are taken after the sweeps, by one thread per m (dsum / dexp_).
template <bool TRI, typename RowPtr>
__device__ __forceinline__ void cphd_esf_chains_f64(const CphdLds& Q, RowPtr rowsP, int rs, int M, int h, int lane, int wave,
                                                    float llam, float lam, LDS_T(double)* dsum, LDS_T(int)* dexp_,
                                                    LDS_T(double)* c_lds, const LDS_T(double)* sh_row, const LDS_T(int)* sh_exp)
{
#pragma clang fp contract(off)
    if (wave >= 4) return;
    // c_a = exp(I1[a]) lambda^(M-1-a) e^-lambda on a common block exponent: the start of the upper chain
    double V = 0.0;
    int kf = -(1 << 28);
    double frac = 0.0;
    if (lane < M) {
        const float Lg = Q.I1[lane] + ((float)(M - 1 - lane) * llam - lam);
        if (Lg > -1e30f) {
            const double t = (double)Lg * 1.4426950408889634;
            const double kc = ceil(t);
            frac = exp2(t - kc);                                     // in (0.5, 1]
            kf = (int)kc;
        }
    }
    int kV = wave_max_i(kf);
    if (kV == -(1 << 28)) kV = 0;                                    // every c_a is zero
    if (kf != -(1 << 28)) { const int d = kf - kV; V = d < -1100 ? 0.0 : __builtin_ldexp(frac, d); }
    const float xv = lane < M ? Q.lxi[lane] : 0.f;
    const bool lower = wave >= 2;
    const int me = wave & 1;
    int m_hi = M - 1, m_lo = h;                                      // this chain's measurements: m_hi .. m_lo
    if (lower) {
        m_hi = h - 1; m_lo = 0;
        // T_h = S_h (*) c: both waves of the pair write the same c (zero-padded to 2 M entries) and read it back
        if (lane < M) { c_lds[lane] = V; c_lds[M + lane] = 0.0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): this wave's own LDS writes have landed
        double acc = 0.0;
        const int deg = M - h;
        if (lane < M)
            for (int b = 0; b <= deg; ++b) acc = __builtin_fma(sh_row[b], c_lds[lane + b], acc);
        V = acc;
        kV += sh_exp[0];
        const int r = wave_max_i(dexp_field(V)) - 1023;
        if (r > -1023) { V = __builtin_ldexp(V, -r); kV += r; }
    }
    int since = 0;
#ifndef PHD_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    // one step of the recursion with root m: V <- V + xi_m shift(V); rescaled every PHD_F64_RENORM steps
    auto step = [&](int m) {
        const double x = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), m));
        V = __builtin_fma(x, lane_down1_z(V), V);
        if (++since == PHD_F64_RENORM) {
            const int r = wave_max_i(dexp_field(V)) - 1023;
            if (r > -1023) { V = __builtin_ldexp(V, -r); kV += r; }
            since = 0;
        }
    };
    // this wave's inner products: m = m_hi - me, m_hi - me - 2, ...; the recursion runs every step.  The per-lane products of
    // EIGHT inner products wait in registers for one transposing reduction (reduce8_over_wave_d): sixteen inner products per
    // wave cost two of those instead of sixteen 6-step butterflies.
    int m = m_hi;
    if (me == 1 && m >= m_lo) { if (m > m_lo) step(m); --m; }
    double prow = (m >= m_lo && lane <= m) ? rowsP[cphd_row_off<TRI>(m, rs) + lane] : 0.0;
    int krow = m >= m_lo ? Q.kp[m] : 0;
    while (m >= m_lo) {
        double pq[8];
        int eq = 0, nq = 0;
        const int mq0 = m;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            pq[q] = 0.0;
            if (m >= m_lo) {                                                            // (uniform)
                pq[q] = prow * V;                                                       // <P_m, T_{m+1}>, lane by lane
                if ((lane >> 3) == q) eq = krow + kV;
                nq = q + 1;
                const int mn = m - 2;                                                   // the row of the next one, a turn ahead
                prow = (mn >= m_lo && lane <= mn) ? rowsP[cphd_row_off<TRI>(mn, rs) + lane] : 0.0;
                krow = mn >= m_lo ? Q.kp[mn] : 0;
                if (m > m_lo) step(m);
                if (m - 1 > m_lo) step(m - 1);
                m -= 2;
            }
        }
        const double tot = reduce8_over_wave_d(pq, lane);                               // lane l: the total of entry l >> 3
        const int q = lane >> 3;
        if ((lane & 7) == 0 && q < nq) { dsum[mq0 - 2 * q] = tot; dexp_[mq0 - 2 * q] = eq; }
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

// eight per-lane maxima reduced over the wave in one transposing pass (the max twin of reduce8_over_wave, phd_pass1.h):
// lane l ends with the wave maximum of a[l >> 3]
__device__ __forceinline__ float max8_over_wave(float (&a)[8], int lane)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 4]), false, false);
        a[i] = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
        a[i] = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    }
    const bool hi8 = (lane & 8) != 0;
    const float keep = hi8 ? a[1] : a[0], send = hi8 ? a[0] : a[1];
    float v = fmaxf(keep, __uint_as_float(dpp_mov<0x128, 0xF>(__float_as_uint(send), __float_as_uint(send))));   // row_ror:8
    v = fmaxf(v, __uint_as_float(dpp_mov<0x141, 0xF>(__float_as_uint(v), __float_as_uint(v))));                  // row_half_mirror
    v = fmaxf(v, __uint_as_float(dpp_mov<0x1B, 0xF>(__float_as_uint(v), __float_as_uint(v))));                   // quad_perm [3,2,1,0]
    v = fmaxf(v, __uint_as_float(dpp_mov<0xB1, 0xF>(__float_as_uint(v), __float_as_uint(v))));                   // quad_perm [1,0,3,2]
    return v;
}

// four per-lane maxima over the wave (the max twin of reduce4_over_wave): lane l ends with the wave maximum of a[l >> 4]
__device__ __forceinline__ float max4_over_wave(float (&a)[4], int lane)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
        a[i] = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    }
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[0]), __float_as_uint(a[1]), false, false);
    float v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    v = fmaxf(v, __uint_as_float(dpp_mov<0x128, 0xF>(__float_as_uint(v), __float_as_uint(v))));                  // row_ror:8
    v = fmaxf(v, __uint_as_float(dpp_mov<0x141, 0xF>(__float_as_uint(v), __float_as_uint(v))));                  // row_half_mirror
    v = fmaxf(v, __uint_as_float(dpp_mov<0x1B, 0xF>(__float_as_uint(v), __float_as_uint(v))));                   // quad_perm [3,2,1,0]
    v = fmaxf(v, __uint_as_float(dpp_mov<0xB1, 0xF>(__float_as_uint(v), __float_as_uint(v))));                   // quad_perm [1,0,3,2]
    return v;
}

// cphd_nsums_fast for cardinality rows of up to 256 entries with EIGHT (or four) j per trip: the two wave reductions of a
// log-sum-exp (maximum, sum) are 2 x 18 instructions per j done one j at a time — more than the terms themselves; transposing
// reductions do eight of each in ~22, four in ~14.  A wave takes the contiguous range [j_lo, j_hi) of j: batches of eight, then
// batches of four (the last one padded: a j outside the range is computed and dropped).  Same sums, another fixed tree.
template <int NB>
__device__ __forceinline__ void cphd_nsums_batch(const CphdLds& Q, int M, int Nmax, int lane, int j0, int j_hi, float lWq)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    float tv[NB][4], mx[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int j = j0 + q;
        mx[q] = LOG0F;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = j + lane + 64 * c;
            tv[q][c] = LOG0F;
            if (n <= Nmax) { tv[q][c] = Q.cnq[n] - Q.lfact[n - j]; mx[q] = fmaxf(mx[q], tv[q][c]); }
        }
    }
    float mxw;                                                       // lane l: the maximum of j0 + (l >> 3) (NB = 8) / (l >> 4) (NB = 4)
    if constexpr (NB == 8) mxw = max8_over_wave(mx, lane); else mxw = max4_over_wave(mx, lane);
    float sacc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const float m = lane_f(mxw, (64 / NB) * q);
        sacc[q] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (j0 + q + lane + 64 * c <= Nmax) sacc[q] += __expf(tv[q][c] - m);
    }
    float sw;
    if constexpr (NB == 8) sw = reduce8_over_wave(sacc, lane); else sw = reduce4_over_wave(sacc, lane);
    if ((lane & (64 / NB - 1)) == 0) {
        const int j = j0 + lane / (64 / NB);
        if (j < j_hi) {
            const float v = (j <= Nmax) ? (safe_log(sw) + mxw) - (float)j * lWq : LOG0F;
            if (j <= M) Q.I0[j] = v;
            if (j >= 1) Q.I1[j - 1] = v;
        }
    }
}

__device__ __forceinline__ void cphd_nsums_fast8(const CphdLds& Q, int M, int Nmax, int lane, int j_lo, int j_hi, float lWq)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    int j0 = j_lo;
    for (; j_hi - j0 >= 8; j0 += 8) cphd_nsums_batch<8>(Q, M, Nmax, lane, j0, j_hi, lWq);
    // what is left (two of the 66 values of a 64-measurement scan over seven waves) one j at a time (batches of four, the last
    // one padded, were measured slower: 1 947 vs 1 975 steps/s)
    for (int j = j0; j < j_hi; ++j) {
        float tv[4];
        float mx = LOG0F;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = j + lane + 64 * c;
            tv[c] = LOG0F;
            if (n <= Nmax) { tv[c] = Q.cnq[n] - Q.lfact[n - j]; mx = fmaxf(mx, tv[c]); }
        }
        mx = wave_max_f(mx);
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
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
                                        int tid, u64* cq, int S_cap, u64* stg = nullptr)
{
#pragma clang fp contract(off)
    // cq (diagnostic instantiation, thread 0): time of [staging + birth cardinality, forward sweep beside the cardinality work,
    // backward sweep + inner products, rest]
#define CQSTAMP(k) do { if (cq && tid == 0) cq[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GSTAMP(k, t) do { if (stg && tid == (t)) stg[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
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
    GSTAMP(29, 0);
    // M <= 64: the sweeps in double (cphd_esf_*_f64); their rows live in the survivor arrays — idle until this block has
    // produced the weights — when those hold M rows of 64 doubles, else in the HBM scratch (row stride MM)
    const bool rows_in_lds = tiles == 1 && 32u * (u32)S_cap >= cphd_rows_bytes(MM);    // (triangular rows; the block's own arrays may follow them)
    LDS_T(double)* const rows_l = (LDS_T(double)*)L.w;
    double* const rows_g = (double*)T_scratch;
    // f64 path (M <= 64): wave 0 runs the forward sweep (parks every row P_m) and then, M / 2 steps, the suffix polynomial S_h of
    // the upper half of the roots, which lets the backward recursion run as two half-length chains side by side afterwards
    const bool f64 = tiles == 1;
    const int hsplit = M >> 1;                                      // (26 of 64 for the lower chain — which also pays the correlation — measured no faster)
    LDS_T(int)* sctr = (LDS_T(int)*)&L.ctr[CTR_WSYNC];
    const bool lin = (size_t)8 * cn_len <= (size_t)32 * MM;
    // (zpart, 32 MM >= 32 M bytes, is idle from the roots to the cardinality update — but for the scaled prior of the predicted
    //  cardinality, see cphd_esf_suffix_f64: [M] inner-product sums | [2 M] c, zero-padded | [M] S_h)
    LDS_T(double)* const dsum = (LDS_T(double)*)L.zpart;
    LDS_T(double)* const c_lds = dsum + M;
    LDS_T(double)* const sh_row = dsum + 3 * M;
    LDS_T(int)* const sh_exp = (LDS_T(int)*)&L.ctr[CTR_NHEAD];        // (a merge counter: free until the merge)
    // (giving the sweep wave, free two thirds into the phase, the last 6 of the 66 n-sums was measured: no gain — 1 967 vs 1 975 steps/s)
    const int j_w0 = 0;
    if (wave == 0) {
        if (f64 && rows_in_lds) cphd_esf_forward_park_f64<true>(Q, rows_l, 64, M, 0, lane);
        else if (f64) cphd_esf_forward_park_f64<false>(Q, rows_g, MM, M, 0, lane);
        else if (tiles == 2) cphd_esf_forward_park<2>(Q, T_scratch, M, lane);
        else cphd_esf_forward_park<4>(Q, T_scratch, M, lane);
        GSTAMP(16, 0);   // forward sweep done
        if (f64) {
            cphd_esf_suffix_f64(Q, sh_row, sh_exp, M, hsplit, lane, sctr, (lin ? 2 : 1) * (PHD_NW - 1));
            GSTAMP(31, 0);   // suffix half-sweep done
        }
    } else {
        int target = 0;
        const int t7 = tid - 64, T7 = PHD_T - 64, w7 = wave - 1, W7 = PHD_NW - 1;
        // predicted cardinality (.bak:518-545): prior (*) Binomial(M, birthWeight).  The log-domain form costs two passes of
        // (two LDS reads, add, max | sub, exp, add) per term, 2 x 65 terms per n: 6 us.  In the LINEAR domain the convolution
        // is one fma per term; doubles, scaled by the prior's maximum, keep everything within e^-700 of that maximum (what
        // lies below is zero for every sum taken from these numbers) — when the idle zpart / I0 / I1 arrays can hold them.
        if (lin) {
            LDS_T(double)* const qd = (LDS_T(double)*)L.zpart;       // [cn_len] exp(prior - max)
            LDS_T(double)* const bd = (LDS_T(double)*)Q.I0;          // [Kb + 1] Binomial pmf (I0 and I1 are written after this phase)
            float qmax = -FLT_MAX;
            for (int n = lane; n <= Nmax; n += 64) qmax = fmaxf(qmax, Q.cnq[n]);   // every wave the same maximum: no exchange
            qmax = wave_max_f(qmax);
            for (int n = t7; n <= Nmax; n += T7) qd[n] = Q.cnq[n] > -1e30f ? exp((double)(Q.cnq[n] - qmax)) : 0.0;
            for (int k = t7; k <= Kb; k += T7) bd[k] = exp((double)Q.cnb[k]);
            waves_sync(sctr, W7, target, lane);
            // Early exit, bit for bit the full sum: the scaled prior is <= 1, so once the binomial term b_k — decreasing beyond
            // its mode — is below 2^-54 of the running sum, b_k q_{n-k} is less than half an ulp of it and the fma returns the sum
            // unchanged, as does every later one.  With birth_weight = 1e-4 that is k = 7 of the 65 terms for a flat prior
            // (13 where the prior is e^-50 below its maximum); a prior that needs all terms still gets them.
            const int kmode = (int)((float)(M + 1) * cfg.birthWeight);                 // mode of Binomial(M, birthWeight)
            for (int n = t7; n <= Nmax; n += T7) {
                const int kmax = n < Kb ? n : Kb;
                double s = 0.0;
                for (int k = 0; k <= kmax; ++k) {
                    const double b = bd[k];
                    if (k > kmode && b < s * 5.551115123125783e-17) break;              // 2^-54
                    s = __builtin_fma(b, qd[n - k], s);
                }
                Q.cnp[n] = s > 0.0 ? log_scaled(s, 0) + qmax : LOG0F;
            }
        } else
        for (int n = t7; n <= Nmax; n += T7) {
            const int kmax = n < Kb ? n : Kb;
            float mx = Q.cnb[0] + Q.cnq[n];
            for (int k = 1; k <= kmax; ++k) mx = fmaxf(mx, Q.cnb[k] + Q.cnq[n - k]);
            float s = 0.f;
            for (int k = 0; k <= kmax; ++k) s += __expf(Q.cnb[k] + Q.cnq[n - k] - mx);
            Q.cnp[n] = safe_log(s) + mx;
        }
        GSTAMP(17, 64);  // predicted cardinality done (first wave of the group)
        waves_sync(sctr, W7, target, lane);
        GSTAMP(18, 64);
        // I_u[j] = log sum_n p(n) P(n,j+u) Wq^(n-j-u) / W1^n.  Since P(n,j+1) Wq^(n-j-1) is the u = 0 term of j+1,
        // I_1[j] = I_0[j+1] (the same floating-point expression): one family J[j] = I_0[j], j = 0..M+1.
        // Wave per j, lanes over n; the terms stay in registers between the max and the sum pass (n <= 1023).
        // (the chunk count is a compile-time constant per cardinality length: max_cardinality 255 needs 4 of the 16)
        if (finite_w) {
            for (int n = t7; n <= Nmax; n += T7) Q.cnq[n] = Q.cnp[n] + Q.lfact[n] + (float)n * (lWq - lW1);   // B_n
            waves_sync(sctr, W7, target, lane);
            GSTAMP(19, 64);  // B_n done
            if (cn_len <= 256) {
                // j = 0 .. M + 1 in contiguous ranges
                const int nj = M + 2 - j_w0;
                const int per = (nj + W7 - 1) / W7;
                const int jl = w7 * per, jh = (jl + per < nj) ? jl + per : nj;
                if (per <= 16) { if (jl < jh) cphd_nsums_fast8(Q, M, Nmax, lane, jl, jh, lWq); }
                else cphd_nsums_fast<4>(Q, M, Nmax, lane, w7, W7, lWq);
            }
            else if (cn_len <= 512) cphd_nsums_fast<8>(Q, M, Nmax, lane, w7, W7, lWq);
            else cphd_nsums_fast<16>(Q, M, Nmax, lane, w7, W7, lWq);
        } else {
            if (cn_len <= 256) cphd_nsums<4>(Q, M, Nmax, lane, w7, W7, lWq, lW1);
            else if (cn_len <= 512) cphd_nsums<8>(Q, M, Nmax, lane, w7, W7, lWq, lW1);
            else cphd_nsums<16>(Q, M, Nmax, lane, w7, W7, lWq, lW1);
        }
        GSTAMP(20, 64);      // n-sums done (first wave of the group)
    }
    __syncthreads();
    CQSTAMP(2);
    GSTAMP(30, 0);
    // ESFs (.bak:1224-1272).  The .bak runs one full recursion per left-out measurement (O(M^3)); here
    //   e(Xi \ m) = P_m (*) S_{m+1}   (ESFs of the roots before and after m), so
    //   <Y1[Z\m],p> = sum_a P_m[a] T_{m+1}[a],  T_{m+1}[a] = sum_b S_{m+1}[b] c_{a+b},  c_j = exp(I1[j]) lambda^(M-1-j) e^-lambda
    // and T obeys the same one-root recursion run backwards, T_m[a] = T_{m+1}[a] + xi_m T_{m+1}[a+1], T_M = c:
    // O(M^2), all terms positive.  The rows P_m[0..m] were parked in HBM scratch by the forward sweep above (M^2 x 8 B
    // per particle — what 288 GB are for; they come back out of L2); the backward sweep carries T in registers and takes
    // one inner product per measurement.  Values span hundreds of decades, so each is a float mantissa with its own
    // integer exponent (m 2^k): align with v_ldexp, renormalise with v_frexp — exact operations around one correctly
    // rounded multiply and add (the oracle does the same).  Wave PHD_FW, idle in the sweep, takes <Y0,p> and <Y1,p>.
    // (the inner products' sums and exponents wait in the — still unused — logZ / zpart arrays for their logarithms)
    LDS_T(int)* const dexp_ = (LDS_T(int)*)L.logZ;
    if (tiles != 1 && wave == PHD_FW) cphd_full_set(Q, M, lane, llam, lam);
    // Round 5: the UPDATED CARDINALITY beside the backward chains.  The chains occupy waves 0-3 (one per SIMD) for ~10 us while
    // waves 4-7 used to wait at the barrier - and the cardinality update (p(n) Y0(n) / <Y0,p>: the full-set ESFs, B_n, <Y0,p>;
    // nothing of the chains) then took the whole workgroup another ~5 us behind it.  Waves 4-7 now run it DURING the chains:
    // wave 7 takes <Y0,p> / <Y1,p> first, the four waves the table of reciprocals and the scaled a_j, then one n per thread -
    // the same loop, the same bits.  (On waves 4 and 5 alone - the SIMDs of the upper chain, which has 3 us of slack - the update
    // itself became the phase's critical path: 13.0 instead of 11.4 us.)  Measured at 4096 x 256 x 64: "weights + cardinality
    // update" 6.3 -> 0.9 us, the chains 9.7 -> 11.4 us (they share their SIMDs now), 2 590 -> 2 650 steps/s.  Its two tables (1/d, exp(a_j - max a): 8 (cn_len + M + 1) bytes) cannot sit where the serial
    // form keeps them (zpart and I0 belong to the chains now): they go behind the block's arrays in the survivor planes, when
    // the block lives there and the planes have the room (4096 x 256 x 64: 2.6 of 9 KB free); else the serial form below.
    u32 cq_off[12];
    const u32 blk_bytes = cphd_lds_layout(cn_len, MM, cq_off);
    const u32 conc_need = align16u(8u * (u32)cn_len) + align16u(8u * (u32)(M + 1));
    const bool lin_upd = finite_w && (size_t)8 * cn_len <= (size_t)32 * MM;
    const bool conc = tiles == 1 && lin_upd && cphd_block_in_planes(S_cap, cn_len, MM) &&
                      32u * (u32)S_cap >= cphd_rows_bytes(MM) + blk_bytes + conc_need;
    if (tiles == 1) {
        if (conc && wave >= 4) {
            LDS_T(double)* const inv = (LDS_T(double)*)((lds_u8)L.w + cphd_rows_bytes(MM) + blk_bytes);   // [cn_len] 1 / d
            LDS_T(double)* const ad = inv + (align16u(8u * (u32)cn_len) >> 3);                            // [M + 1] exp(a_j - max a)
            LDS_T(int)* const cctr = (LDS_T(int)*)&L.ctr[CTR_WSYNC2];
            const int t4 = tid - 4 * 64, T4 = PHD_T - 4 * 64;
            int target = 0;
            if (wave == PHD_NW - 1) cphd_full_set(Q, M, lane, llam, lam);
            else {
                for (int j = t4; j <= M; j += T4 - 64) Q.cnb[j] = Q.efull[j] + ((float)(M - j) * llam - lam) - (float)j * lWq;
                for (int d = t4; d <= Nmax; d += T4 - 64) inv[d] = d ? 1.0 / (double)d : 0.0;
            }
            waves_sync(cctr, 4, target, lane);
            float amax = -FLT_MAX;
            for (int j = lane; j <= M; j += 64) amax = fmaxf(amax, Q.cnb[j]);     // every wave the same maximum
            amax = wave_max_f(amax);
            for (int j = t4; j <= M; j += T4) ad[j] = Q.cnb[j] > -1e30f ? exp((double)(Q.cnb[j] - amax)) : 0.0;
            waves_sync(cctr, 4, target, lane);
            const float lY0c = Q.scal[CQ_LY0];
            for (int n = t4; n <= Nmax; n += T4) {
                const int jm = n < M ? n : M;
                double r = 1.0, sd = ad[jm];
                for (int j = jm - 1; j >= 0; --j) {
                    r *= inv[n - j];
                    if (r < sd * 5.551115123125783e-17) break;                             // 2^-54
                    sd = __builtin_fma(ad[j], r, sd);
                }
                cn_out[n] = sd > 0.0 ? Q.cnq[n] + ((log_scaled(sd, 0) + amax) - Q.lfact[n - jm]) - lY0c : LOG0F;
            }
            GSTAMP(23, 256);     // the concurrent cardinality update done (first of its waves)
        } else {
            if (rows_in_lds) cphd_esf_chains_f64<true>(Q, (const LDS_T(double)*)rows_l, 64, M, hsplit, lane, wave, llam, lam, dsum, dexp_, c_lds, sh_row, sh_exp);
            else cphd_esf_chains_f64<false>(Q, (const double*)rows_g, MM, M, hsplit, lane, wave, llam, lam, dsum, dexp_, c_lds, sh_row, sh_exp);
            GSTAMP(21, 0);       // wave 0's share of the backward sweep done
            GSTAMP(22, 192);     // wave 3's
            if (wave == PHD_NW - 1) cphd_full_set(Q, M, lane, llam, lam);   // (after its share of the sweep: every wave takes part in it)
        }
    }
    else if (tiles == 2) cphd_esf_backward_dot<2>(Q, T_scratch, M, lane, wave, llam, lam);
    else cphd_esf_backward_dot<4>(Q, T_scratch, M, lane, wave, llam, lam);
    __syncthreads();
    CQSTAMP(3);
    const float lY0 = Q.scal[CQ_LY0];
    if (tiles == 1) {
        for (int m = tid; m < M; m += PHD_T) Q.lD[m] = log_scaled(dsum[m], dexp_[m]);       // log <Y1[Z \ m], p>
        __syncthreads();                                                                  // (logZ aliases the exponents)
    }
    for (int m = tid; m < M; m += PHD_T) L.logZ[m] = -((llam - lkap) + Q.lD[m] - lY0);      // .bak:1434-1437
    if (tid == 0) Q.scal[CQ_R1] = expf(Q.scal[CQ_LY1] - lY0);                               // .bak:1452-1455
    // updated cardinality (.bak:1409-1411): p(n) Y0(n) / <Y0,p>.  With B_n as above and a_j = log e_j + (M-j) log lambda
    // - lambda - j log Wq (in the cnb array, free by now) the term is B_n + a_j - log (n-j)!: the sum over j costs two LDS
    // reads and a subtraction per term
    if (conc) {
        // (done beside the chains, above)
    } else if (finite_w && (size_t)8 * cn_len <= (size_t)32 * MM) {
        // the same sum in the LINEAR domain (doubles): with a~_j = exp(a_j - max a) and the factorial ratio carried as a
        // running product, sum_j exp(a_j - log (n-j)!) = exp(max a) / (n-jm)! * sum_j a~_j r_j,  r_jm = 1, r_{j-1} = r_j / (n-j+1)
        // (jm = min(n, M): the largest term's factorial is the unit, every other ratio is below one) — a multiply and an fma
        // per term instead of two passes with an exponential; 1/d comes from a table in the (idle again) zpart array
        LDS_T(double)* const inv = (LDS_T(double)*)L.zpart;          // [cn_len] 1 / d
        LDS_T(double)* const ad = (LDS_T(double)*)Q.I0;              // [M + 1] exp(a_j - max a) (I0 / I1 are consumed)
        for (int j = tid; j <= M; j += PHD_T) Q.cnb[j] = Q.efull[j] + ((float)(M - j) * llam - lam) - (float)j * lWq;
        for (int d = tid; d <= Nmax; d += PHD_T) inv[d] = d ? 1.0 / (double)d : 0.0;
        __syncthreads();
        float amax = -FLT_MAX;
        for (int j = lane; j <= M; j += 64) amax = fmaxf(amax, Q.cnb[j]);     // every wave the same maximum
        amax = wave_max_f(amax);
        for (int j = tid; j <= M; j += PHD_T) ad[j] = Q.cnb[j] > -1e30f ? exp((double)(Q.cnb[j] - amax)) : 0.0;
        __syncthreads();
        // (two threads per n — upper and lower half of the j range, joined by r_mid — was measured slower: 6.3 vs 4.7 us for the
        //  phase; with the other resident workgroup on the SIMDs the loop is bound by issue, not by the idle half of the threads)
        // (early exit, bit for bit the full sum: a~_j <= 1 and r only shrinks — by 1 / (n - j + 1) per step — so once r is below
        //  2^-54 of the running sum no later term can change it: ~8 of the 64 terms at n = 200, ~19 at n <= M)
        for (int n = tid; n <= Nmax; n += PHD_T) {
            const int jm = n < M ? n : M;
            double r = 1.0, sd = ad[jm];
            for (int j = jm - 1; j >= 0; --j) {
                r *= inv[n - j];
                if (r < sd * 5.551115123125783e-17) break;                             // 2^-54
                sd = __builtin_fma(ad[j], r, sd);
            }
            cn_out[n] = sd > 0.0 ? Q.cnq[n] + ((log_scaled(sd, 0) + amax) - Q.lfact[n - jm]) - lY0 : LOG0F;
        }
    } else if (finite_w) {
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
    // the bucket counts of the merge's counting sort (phd_lds.h): the `tr` plane held sweep rows until now — and, with a long
    // cardinality distribution, may hold part of this block's own arrays, which the loops above still read
    if (cphd_block_in_planes(S_cap, cn_len, MM)) __syncthreads();   // (uniform)
    for (int b = tid; b < S_cap; b += PHD_T) ((LDS_T(u32)*)L.tr)[b] = 0u;
    __syncthreads();
    CQSTAMP(4);
#undef CQSTAMP
#undef GSTAMP
}

} // namespace phd
