// phd_kernels.hip — gfx950 (CDNA4) kernels of the GM-PHD-SLAM hot path.
//
// One workgroup (256 threads = 4 wave64) owns one particle for the whole measurement update:
//
//   classify map (in range / nearly in range / out)        reference: computeInRangeKernel  src/phdfilter.cu:1279-1358
//   per-feature EKF terms -> LDS (SoA)                      reference: preUpdateSynthKernel  src/phdfilter.cu:1824-1925
//   pass 1: per-measurement normalisers  Z_m                reference: phdUpdateKernel       src/phdfilter.cu:2190-2223
//   pass 2: final weights, prune BEFORE store, survivors -> LDS      src/phdfilter.cu:2226-2245,2307-2319 + pruneMap :3120-3174
//   greedy merge of the survivors, entirely in LDS          reference: phdUpdateMergeKernel  src/phdfilter.cu:2707-2898
//   merged map + untouched out-of-range features -> HBM     reference: mergeAndCopyMaps      src/phdfilter.cu:3304-3318
//
// The reference materialises all G(M+1)+M update components per particle in global memory
// twice (1.9 GB at 4096x256x64) and makes 3 global passes per merged Gaussian.  Here the update
// components never leave registers unless they survive pruning, survivors live in LDS, and HBM
// sees only the map read, the map write and a few bytes per particle.
//
// Lane mapping of the two (feature x measurement) passes: lanes <-> measurements, waves <->
// feature slices, so the per-measurement sum over features is a private accumulator — no
// cross-lane reduction in the inner loop (one shuffle tree per wave at the very end).
//
// Merge: exact greedy semantics (seed = max weight, ties -> lowest slab index; absorb d < T),
// re-organised so that the sequential part is 64 candidates per round instead of one:
//   sort survivors by (weight desc, slab index asc)  [bitonic, LDS]
//   per round: 64x64 closeness matrix of the window -> seeds by a scalar bit-mask recurrence ->
//              every later survivor is assigned to the first seed (in order) it is close to
//   sort by (seed, position) -> contiguous clusters -> one lane per cluster does the moment
//   matching sequentially in (weight desc) order — the order gm_reduce.cpp:103-118 uses.
// A conservative trace bound (d >= 2|dm|^2/(tr Pa + tr Pb)) rejects far pairs before the exact
// 2x2 inverse; it never rejects a pair the exact test would accept (1 % guard band).
//
// No CUDA compatibility layer, no Thrust/hipCUB, wave64 only.

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "phd_device.h"

namespace phd {

#define PHD_T 256
#define PHD_NW 4
#define NEAR_U_BASE 0x40000000

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;
#define LDS_T(T) __attribute__((address_space(3))) T

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float safe_log(float x) { return (x <= 0.f) ? -FLT_MAX : logf(x); } // device_math.cuh:9-16

// wrapAngle, src/device_math.cuh:241-251.  fmod(a,2pi) == a exactly when |a| < 2pi; the
// reference compares against the double M_PI (rem > M_PI  <=>  rem >= float(pi)) and subtracts
// the double 2*M_PI: rem - 2pi = (rem - A) - B with A = float(2pi) (exact by Sterbenz), B = 2pi - A.
__device__ __forceinline__ float wrap_angle(float a)
{
    const float TWO_PI_F = 6.2831855f;
    const float PI_F = 3.14159274f;
    const float B = -1.7484555e-7f; // 2*pi - float(2*pi)
    float rem = (fabsf(a) < TWO_PI_F) ? a : fmodf(a, TWO_PI_F);
    if (rem >= PI_F) rem = (rem - TWO_PI_F) - B;
    else if (rem <= -PI_F) rem = (rem + TWO_PI_F) + B;
    return rem;
}

__device__ __forceinline__ u64 lanemask_lt()
{
    u32 lane = __lane_id();
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide sum with a fixed reduction tree (deterministic); scratch: PHD_NW floats
__device__ __forceinline__ float block_sum(float v, LDS_T(float)* scratch, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < PHD_NW; ++w) r += scratch[w];
    return r;
}

__device__ __forceinline__ u32 orderable(float w)
{
    u32 b = __float_as_uint(w);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ------------------------------------------------------------------------------------------
// EKF terms of one in-range feature (src/phdfilter.cu:1841-1894), fp32, reference expression order
// ------------------------------------------------------------------------------------------
struct EkfTerms {
    float r, b, pd;
    float s00, s12, s11; // S = Sigma^-1: S[0], S[1]+S[2], S[3]
    float det;
    float K0, K1, K2, K3;
    float J0, J1, J2, J3;
};

__device__ __forceinline__ void ekf_terms(float mx, float my, float pxx, float pxy, float pyy,
                                          const phd_pose& pose, const DevConfig& cfg, EkfTerms& t)
{
    float dx = mx - pose.px;
    float dy = my - pose.py;
    float r2 = dx * dx + dy * dy;
    float r = sqrtf(r2);
    float bearing = wrap_angle(atan2f(dy, dx) - pose.ptheta);
    t.r = r;
    t.b = bearing;
    t.pd = (r <= cfg.maxRange && fabsf(bearing) <= cfg.maxBearing) ? cfg.pd : 0.f; // :1848-1850
    float J0 = dx / r, J2 = dy / r, J1 = -dy / r2, J3 = dx / r2;                  // :1854-1858
    const float P0 = pxx, P1 = pxy, P2 = pxy, P3 = pyy;
    float sg0 = (P0 * J0 + J2 * P1) * J0 + (J0 * P2 + P3 * J2) * J2 + cfg.stdRange * cfg.stdRange;
    float sg1 = (P0 * J1 + J3 * P1) * J0 + (J1 * P2 + P3 * J3) * J2;
    float sg2 = (P0 * J0 + J2 * P1) * J1 + (J0 * P2 + P3 * J2) * J3;
    float sg3 = (P0 * J1 + J3 * P1) * J1 + (J1 * P2 + P3 * J3) * J3 + cfg.stdBearing * cfg.stdBearing;
    sg1 = (sg1 + sg2) * 0.5f;                                                    // :1871-1872
    sg2 = sg1;
    float det = sg0 * sg3 - sg1 * sg2;                                           // :1874
    float S0 = sg3 / det, S1 = -sg1 / det, S2 = -sg2 / det, S3 = sg0 / det;      // :1877-1881
    t.det = det;
    t.s00 = S0;
    t.s12 = S1 + S2;
    t.s11 = S3;
    t.K0 = S0 * (P0 * J0 + P2 * J2) + S1 * (P0 * J1 + P2 * J3);                  // :1884-1888
    t.K1 = S0 * (P1 * J0 + P3 * J2) + S1 * (P1 * J1 + P3 * J3);
    t.K2 = S2 * (P0 * J0 + P2 * J2) + S3 * (P0 * J1 + P2 * J3);
    t.K3 = S2 * (P1 * J0 + P3 * J2) + S3 * (P1 * J1 + P3 * J3);
    t.J0 = J0; t.J1 = J1; t.J2 = J2; t.J3 = J3;
}

// Joseph-form covariance (src/phdfilter.cu:1891-1894); returns the symmetric part
__device__ __forceinline__ void joseph_cov(const EkfTerms& t, float pxx, float pxy, float pyy,
                                           const DevConfig& cfg, float& oxx, float& oxy, float& oyy)
{
    const float P0 = pxx, P1 = pxy, P2 = pxy, P3 = pyy;
    const float sr = cfg.stdRange, sb = cfg.stdBearing;
    float a00 = 1 - t.K0 * t.J0 - t.K2 * t.J1;
    float a01 = -t.K0 * t.J2 - t.K2 * t.J3;
    float a10 = -t.K1 * t.J0 - t.K3 * t.J1;
    float a11 = 1 - t.K1 * t.J2 - t.K3 * t.J3;
    float c0 = (a00 * P0 + a01 * P1) * a00 + (a00 * P2 + a01 * P3) * a01 + t.K0 * t.K0 * sr * sr + t.K2 * t.K2 * sb * sb;
    float c2 = (a00 * P0 + a01 * P1) * a10 + (a00 * P2 + a01 * P3) * a11 + t.K0 * sr * sr * t.K1 + t.K2 * sb * sb * t.K3;
    float c1 = (a10 * P0 + a11 * P1) * a00 + (a10 * P2 + a11 * P3) * a01 + t.K0 * sr * sr * t.K1 + t.K2 * sb * sb * t.K3;
    float c3 = (a10 * P0 + a11 * P1) * a10 + (a10 * P2 + a11 * P3) * a11 + t.K1 * t.K1 * sr * sr + t.K3 * t.K3 * sb * sb;
    oxx = c0;
    oxy = (c1 + c2) * 0.5f; // the merge symmetrises anyway (force_symmetric_covariance, device_math.cuh:710-725)
    oyy = c3;
}

// ------------------------------------------------------------------------------------------
// exact distances — evaluated in the reference's operation order with FMA contraction off, so
// the merge reproduces the CPU oracle bit for bit on identical inputs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float mahal_dist(float amx, float amy, float axx, float axy, float ayy,
                                            float bmx, float bmy, float bxx, float bxy, float byy)
{
#pragma clang fp contract(off)
    // src/device_math.cuh:308-325 (+ invert_matrix2 :62-69)
    float s0 = (axx + bxx) * 0.5f;
    float s1 = (axy + bxy) * 0.5f;
    float s2 = s1;
    float s3 = (ayy + byy) * 0.5f;
    float det = s0 * s3 - s2 * s1;
    float i0v = s3 / det;
    float i1v = -s1 / det;
    float i2v = -s2 / det;
    float i3v = s0 / det;
    float d0 = amx - bmx;
    float d1 = amy - bmy;
    return d0 * d0 * i0v + d0 * d1 * (i1v + i2v) + d1 * d1 * i3v;
}

__device__ __forceinline__ float hellinger_dist(float amx, float amy, float axx, float axy, float ayy,
                                                float bmx, float bmy, float bxx, float bxy, float byy)
{
#pragma clang fp contract(off)
    // src/device_math.cuh:373-413
    float d0 = amx - bmx, d1 = amy - bmy;
    float g0 = axx + bxx, g1 = axy + bxy, g2 = g1, g3 = ayy + byy;
    float det = g0 * g3 - g2 * g1;
    float v0 = 1.f, v1 = 0.f, v2 = 0.f, v3 = 1.f;
    if (det > FLT_MIN) { v0 = g3 / det; v1 = -g1 / det; v2 = -g2 / det; v3 = g0 / det; }
    float eps = (float)(-0.25 * (double)(d0 * d0 * v0 + d0 * d1 * (v1 + v2) + d1 * d1 * v3));
    det = det / 4;
    float dist = 1 / det;
    float q0 = axx * bxx + axy * bxy;
    float q1 = axy * bxx + ayy * bxy;
    float q2 = axx * bxy + axy * byy;
    float q3 = axy * bxy + ayy * byy;
    det = q0 * q3 - q2 * q1;
    dist *= sqrtf(det);
    dist = 1 - sqrtf(dist) * expf(eps);
    return dist;
}

// ------------------------------------------------------------------------------------------
// LDS layout of the update+merge kernel (dynamic shared memory, carved on the host by
// phd_update_lds_bytes()).  S = survivor capacity (power of two), C = map capacity, MM = max meas.
// ------------------------------------------------------------------------------------------
// LDS pointers carry their address space so every access is a ds_* instruction (a generic
// pointer would compile to flat_* and keep the pointer table in scratch).
typedef LDS_T(float)* lds_f32;
typedef LDS_T(int)* lds_i32;
typedef LDS_T(u32)* lds_u32;
typedef LDS_T(u16)* lds_u16;
typedef LDS_T(unsigned char)* lds_u8;

struct LdsOffsets {
    u32 w, mx, my, xx, xy, yy, tr, u;
    u32 alias;      // start of the aliased region
    u32 out_idx, z_r, z_b, logZ, zpart, zok, part, red, ctr;
    u32 total;
};

__host__ __device__ __forceinline__ u32 align16u(u32 x) { return (x + 15u) & ~15u; }

__host__ __device__ __forceinline__ LdsOffsets lds_offsets(int S, int C, int MM)
{
    LdsOffsets o;
    u32 p = 0;
    const u32 sv = align16u(4u * (u32)S);
    o.w = p; p += sv; o.mx = p; p += sv; o.my = p; p += sv; o.xx = p; p += sv;
    o.xy = p; p += sv; o.yy = p; p += sv; o.tr = p; p += sv; o.u = p; p += sv;
    o.alias = p;
    const u32 feat = 6u * align16u(4u * (u32)C) + align16u(2u * (u32)C);
    const u32 sort1 = 3u * sv;
    const u32 sort2 = sv + align16u(4u * (u32)(S + 1));
    u32 amax = feat > sort1 ? feat : sort1;
    amax = amax > sort2 ? amax : sort2;
    p += amax;
    o.out_idx = p; p += align16u(2u * (u32)C);
    o.z_r = p; p += align16u(4u * (u32)MM);
    o.z_b = p; p += align16u(4u * (u32)MM);
    o.logZ = p; p += align16u(4u * (u32)MM);
    o.zpart = p; p += align16u(16u * (u32)MM);
    o.zok = p; p += align16u(4u * (u32)MM);
    o.part = p; p += 4u * 2u * 4u * 64u;
    o.red = p; p += align16u(4u * (PHD_NW + 4));
    o.ctr = p; p += 4u * 16u;
    o.total = p;
    return o;
}

struct Lds {
    // survivors (SoA), S entries each
    lds_f32 w, mx, my, xx, xy, yy, tr;
    lds_i32 u; // slab index (sort tie-break); after the sort: cluster assignment
    // aliased region
    lds_f32 f_r, f_b, f_s00, f_s12, f_s11, f_lwb; // per in-range feature, C entries
    lds_u16 f_idx;                                // map index of in-range feature j
    lds_u32 khi, klo, pay;                        // sort 1
    lds_u32 key2;                                 // sort 2
    lds_i32 seg;                                  // cluster starts, S+1
    // not aliased
    lds_u16 out_idx;                  // C
    lds_f32 z_r, z_b, logZ, zpart;    // MM, MM, MM, 4*MM
    lds_u32 zok;                      // MM
    lds_u32 part;                     // 2*4*64 row parts of the window closeness matrix
    lds_f32 red;                      // PHD_NW + 4
    lds_i32 ctr;                      // 16 counters
};

__device__ __forceinline__ Lds lds_carve(lds_u8 base, int S, int C, int MM)
{
    const LdsOffsets o = lds_offsets(S, C, MM);
    Lds L;
    L.w = (lds_f32)(base + o.w); L.mx = (lds_f32)(base + o.mx); L.my = (lds_f32)(base + o.my);
    L.xx = (lds_f32)(base + o.xx); L.xy = (lds_f32)(base + o.xy); L.yy = (lds_f32)(base + o.yy);
    L.tr = (lds_f32)(base + o.tr); L.u = (lds_i32)(base + o.u);
    const u32 fc = align16u(4u * (u32)C);
    u32 f = o.alias;
    L.f_r = (lds_f32)(base + f); f += fc;
    L.f_b = (lds_f32)(base + f); f += fc;
    L.f_s00 = (lds_f32)(base + f); f += fc;
    L.f_s12 = (lds_f32)(base + f); f += fc;
    L.f_s11 = (lds_f32)(base + f); f += fc;
    L.f_lwb = (lds_f32)(base + f); f += fc;
    L.f_idx = (lds_u16)(base + f);
    const u32 sv = align16u(4u * (u32)S);
    L.khi = (lds_u32)(base + o.alias);
    L.klo = (lds_u32)(base + o.alias + sv);
    L.pay = (lds_u32)(base + o.alias + 2u * sv);
    L.key2 = (lds_u32)(base + o.alias);
    L.seg = (lds_i32)(base + o.alias + sv);
    L.out_idx = (lds_u16)(base + o.out_idx);
    L.z_r = (lds_f32)(base + o.z_r); L.z_b = (lds_f32)(base + o.z_b); L.logZ = (lds_f32)(base + o.logZ);
    L.zpart = (lds_f32)(base + o.zpart); L.zok = (lds_u32)(base + o.zok);
    L.part = (lds_u32)(base + o.part);
    L.red = (lds_f32)(base + o.red);
    L.ctr = (lds_i32)(base + o.ctr);
    return L;
}

size_t update_lds_bytes(int S, int C, int MM) { return lds_offsets(S, C, MM).total; }

enum { CTR_NSURV = 0, CTR_NIN = 1, CTR_NOUT = 2, CTR_OVERFLOW = 3, CTR_KOUT = 4, CTR_NHEAD = 5, CTR_TMP = 6 /* ..+PHD_NW*3 */ };

// append one survivor; slot allocation is wave-aggregated (one LDS atomic per wave per call site)
__device__ __forceinline__ int alloc_slots(bool keep, lds_i32 ctr)
{
    u64 bal = __ballot(keep);
    int slot = -1;
    if (bal) {
        int base = 0;
        if ((u64)(1ull << __lane_id()) == (bal & (~bal + 1))) base = atomicAdd((int*)&ctr[CTR_NSURV], __popcll(bal));
        int leader = __builtin_ctzll(bal);
        base = __shfl(base, leader);
        slot = base + __popcll(bal & lanemask_lt());
    }
    return slot;
}

__device__ __forceinline__ void store_survivor(const Lds& L, int slot, int S, float w, float mx, float my, float xx,
                                               float xy, float yy, int u)
{
    if (slot < S) {
        L.w[slot] = w; L.mx[slot] = mx; L.my[slot] = my;
        L.xx[slot] = xx; L.xy[slot] = xy; L.yy[slot] = yy;
        L.u[slot] = u;
    } else {
        L.ctr[CTR_OVERFLOW] = 1;
    }
}

// ------------------------------------------------------------------------------------------
// bitonic sorts in LDS (n a power of two)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sort_desc_64(lds_u32 khi, lds_u32 klo, lds_u32 pay, int n, int tid)
{
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += PHD_T) {
                int l = i ^ j;
                if (l > i) {
                    u64 a = ((u64)khi[i] << 32) | klo[i];
                    u64 b = ((u64)khi[l] << 32) | klo[l];
                    bool desc = ((i & k) == 0);
                    bool sw = desc ? (a < b) : (a > b);
                    if (sw) {
                        u32 pa = pay[i], pb = pay[l];
                        khi[i] = (u32)(b >> 32); klo[i] = (u32)b; pay[i] = pb;
                        khi[l] = (u32)(a >> 32); klo[l] = (u32)a; pay[l] = pa;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void sort_asc_32(lds_u32 key, int n, int tid)
{
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += PHD_T) {
                int l = i ^ j;
                if (l > i) {
                    u32 a = key[i], b = key[l];
                    bool asc = ((i & k) == 0);
                    bool sw = asc ? (a > b) : (a < b);
                    if (sw) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// closeness of survivor i to seed s (sorted arrays).  Conservative pre-test then exact test.
template <bool HELLINGER>
__device__ __forceinline__ bool is_close(const Lds& L, int s, float smx, float smy, float sxx, float sxy, float syy,
                                         float str, float emx, float emy, float exx, float exy, float eyy, float etr,
                                         float T, float Tpre)
{
    (void)s;
    if (!HELLINGER) {
        float dx = smx - emx, dy = smy - emy;
        float d2 = dx * dx + dy * dy;
        // d >= 2|dm|^2 / (tr Ps + tr Pe) for SPD covariances: far if 2 d2 >= 1.01 T (trs + tre)
        if (2.f * d2 >= Tpre * (str + etr)) return false;
        return mahal_dist(smx, smy, sxx, sxy, syy, emx, emy, exx, exy, eyy) < T;
    } else {
        return hellinger_dist(smx, smy, sxx, sxy, syy, emx, emy, exx, exy, eyy) < T;
    }
}

// ------------------------------------------------------------------------------------------
// the greedy merge on the survivors held in LDS; writes the merged map to the output slab.
// Returns (in ctr[CTR_KOUT]) the number of merged Gaussians.
// ------------------------------------------------------------------------------------------
template <bool HELLINGER>
__device__ __forceinline__ void merge_in_lds(const Lds& L, int S_cap, int n_surv, const DevConfig& cfg, float* __restrict__ out_slab,
                             int cap, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const float T = cfg.minSeparation;
    // the trace bound needs T > 0; Tpre = +inf-safe guard band
    const float Tpre = (T > 0.f) ? T * 1.01f : -1.f; // T <= 0: "2 d2 >= -(..)" is always true -> far, like the exact test
    const int S = n_surv;
    if (tid == 0) { L.ctr[CTR_KOUT] = 0; L.ctr[CTR_NHEAD] = 0; }
    if (S == 0) { __syncthreads(); return; }

    // ---- sort 1: (weight desc, slab index asc) ------------------------------------------------
    int n_pad = 2;
    while (n_pad < S) n_pad <<= 1;
    for (int i = tid; i < n_pad; i += PHD_T) {
        if (i < S) {
            L.khi[i] = orderable(L.w[i]);
            L.klo[i] = 0xFFFFFFFFu - (u32)L.u[i];
            L.pay[i] = (u32)i;
        } else {
            L.khi[i] = 0; L.klo[i] = 0; L.pay[i] = 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    sort_desc_64(L.khi, L.klo, L.pay, n_pad, tid);
    // permute the survivor arrays into sorted order through registers (8 elements per thread per trip)
    for (int i0 = 0; i0 < S; i0 += PHD_T * 8) {
        float rw[8], rmx[8], rmy[8], rxx[8], rxy[8], ryy[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int i = i0 + e * PHD_T + tid;
            if (i < S) {
                int s = (int)L.pay[i];
                rw[e] = L.w[s]; rmx[e] = L.mx[s]; rmy[e] = L.my[s];
                rxx[e] = L.xx[s]; rxy[e] = L.xy[s]; ryy[e] = L.yy[s];
            }
        }
        // a trip only touches positions [i0, i0+2048) as destination but reads arbitrary sources:
        // all reads of ALL trips must precede any write, so S_cap <= 2048 is required (host checks).
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int i = i0 + e * PHD_T + tid;
            if (i < S) {
                L.w[i] = rw[e]; L.mx[i] = rmx[e]; L.my[i] = rmy[e];
                L.xx[i] = rxx[e]; L.xy[i] = rxy[e]; L.yy[i] = ryy[e];
                bool spd = (rxx[e] > 0.f) && (ryy[e] > 0.f) && (rxx[e] * ryy[e] - rxy[e] * rxy[e] > 0.f);
                L.tr[i] = spd ? (rxx[e] + ryy[e]) : INFINITY;
                L.u[i] = -1; // unassigned
            }
        }
    }
    __syncthreads();

    // ---- rounds: 64 candidates at a time --------------------------------------------------------
    lds_i32 assign = L.u;
    int round = 0;
    for (int base = 0; base < S; base += 64, ++round) {
        lds_u32 part = L.part + (round & 1) * (4 * 64);
        // (1) closeness matrix rows: lane = candidate k, wave = column block [16*wave, 16*wave+16)
        {
            const int k = lane;
            const int ik = base + k;
            u32 bits = 0;
            const bool kvalid = (ik < S) && (assign[ik] < 0);
            float kmx = 0, kmy = 0, kxx = 0, kxy = 0, kyy = 0, ktr = 0;
            if (kvalid) { kmx = L.mx[ik]; kmy = L.my[ik]; kxx = L.xx[ik]; kxy = L.xy[ik]; kyy = L.yy[ik]; ktr = L.tr[ik]; }
            for (int c = 0; c < 16; ++c) {
                const int l = wave * 16 + c;
                const int il = base + l;
                if (il >= S) break;              // uniform
                if (assign[il] >= 0) continue;   // uniform (broadcast read): merged columns cannot be seeds
                const float lmx = L.mx[il], lmy = L.my[il], lxx = L.xx[il], lxy = L.xy[il], lyy = L.yy[il], ltr = L.tr[il];
                if (kvalid && l < k) {
                    if (is_close<HELLINGER>(L, il, lmx, lmy, lxx, lxy, lyy, ltr, kmx, kmy, kxx, kxy, kyy, ktr, T, Tpre))
                        bits |= (1u << c);
                }
            }
            part[wave * 64 + k] = bits;
        }
        __syncthreads();
        // (2) seeds of this window (every wave computes the same uniform mask)
        u64 seeds = 0;
        {
            const int ik = base + lane;
            const bool unm = (ik < S) && (assign[ik] < 0);
            u32 lo = (part[0 * 64 + lane] & 0xFFFFu) | (part[1 * 64 + lane] << 16);
            u32 hi = (part[2 * 64 + lane] & 0xFFFFu) | (part[3 * 64 + lane] << 16);
            const u64 unmerged = __ballot(unm);
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                u32 rlo = __builtin_amdgcn_readlane(lo, k);
                u32 rhi = __builtin_amdgcn_readlane(hi, k);
                u64 rk = ((u64)rhi << 32) | rlo;
                if (((unmerged >> k) & 1ull) && !(rk & seeds)) seeds |= (1ull << k);
            }
            if (wave == 0 && unm) {
                u64 row = ((u64)hi << 32) | lo;
                int owner = ((seeds >> lane) & 1ull) ? lane : __builtin_ctzll(row & seeds);
                assign[ik] = base + owner;
            }
        }
        // (3) survivors beyond the window: first seed (in order) that is close
        if (seeds) {
            for (int i = base + 64 + tid; i < S; i += PHD_T) {
                if (assign[i] >= 0) continue;
                const float emx = L.mx[i], emy = L.my[i], exx = L.xx[i], exy = L.xy[i], eyy = L.yy[i], etr = L.tr[i];
                u64 rem = seeds;
                int found = -1;
                while (rem) {
                    const int s = __builtin_ctzll(rem);
                    rem &= rem - 1;
                    const int is = base + s;
                    if (is_close<HELLINGER>(L, is, L.mx[is], L.my[is], L.xx[is], L.xy[is], L.yy[is], L.tr[is],
                                            emx, emy, exx, exy, eyy, etr, T, Tpre)) {
                        found = is;
                        break;
                    }
                }
                if (found >= 0) assign[i] = found;
            }
        }
        // no barrier here: the next round writes the other half of `part`, and its barrier
        // (after step 1) orders this round's assign[] writes before they are read in step 2.
        // Step 1 of the next round reads assign[] of its own window: elements written in step 3
        // by other waves -> needs ordering.
        __syncthreads();
    }

    // ---- sort 2: group by seed, members in sorted-position order ---------------------------------
    for (int i = tid; i < n_pad; i += PHD_T) L.key2[i] = (i < S) ? (((u32)assign[i] << 16) | (u32)i) : 0xFFFFFFFFu;
    __syncthreads();
    sort_asc_32(L.key2, n_pad, tid);
    // cluster heads -> seg[]
    {
        int running = 0;
        for (int i0 = 0; i0 < S; i0 += PHD_T) {
            const int i = i0 + tid;
            bool head = false;
            if (i < S) head = (i == 0) || ((L.key2[i] >> 16) != (L.key2[i - 1] >> 16));
            const u64 bal = __ballot(head);
            if (lane == 0) L.ctr[CTR_TMP + wave] = __popcll(bal);
            __syncthreads();
            int off = running;
            for (int w = 0; w < wave; ++w) off += L.ctr[CTR_TMP + w];
            int total = 0;
            for (int w = 0; w < PHD_NW; ++w) total += L.ctr[CTR_TMP + w];
            if (head) L.seg[off + __popcll(bal & lanemask_lt())] = i;
            running += total;
            __syncthreads();
        }
        if (tid == 0) { L.seg[running] = S; L.ctr[CTR_NHEAD] = running; L.ctr[CTR_KOUT] = 0x7FFFFFFF; }
        __syncthreads();
    }
    const int n_clusters = L.ctr[CTR_NHEAD];

    // ---- moment matching: one lane per cluster, sequential in (weight desc) order ----------------
    // (two trips over the lanes' clusters: first find where the reference's loop would stop)
    for (int c0 = 0; c0 < n_clusters; c0 += PHD_T) {
#pragma clang fp contract(off)
        const int c = c0 + tid;
        if (c < n_clusters) {
            const int b = L.seg[c], e = L.seg[c + 1];
            const int sp = (int)(L.key2[b] & 0xFFFFu); // the seed (first in sorted order)
            const float smx = L.mx[sp], smy = L.my[sp], sxx = L.xx[sp], sxy = L.xy[sp], syy = L.yy[sp];
            float dself = HELLINGER ? hellinger_dist(smx, smy, sxx, sxy, syy, smx, smy, sxx, sxy, syy)
                                    : mahal_dist(smx, smy, sxx, sxy, syy, smx, smy, sxx, sxy, syy);
            const bool selfok = dself < T;
            const int b0 = selfok ? b : b + 1; // a seed that is not close to itself is not in its own cluster
            float W = 0.f, sx = 0.f, sy = 0.f;
            for (int i = b0; i < e; ++i) {
                const int p = (int)(L.key2[i] & 0xFFFFu);
                const float w = L.w[p];
                W += w;
                sx += w * L.mx[p];
                sy += w * L.my[p];
            }
            // reference loop: W == 0 -> break (src/phdfilter.cu:2821); a seed left unmerged is re-picked
            // and then yields W == 0
            int stop_at = 0x7FFFFFFF;
            if (W == 0.f) stop_at = c;
            else if (!selfok) stop_at = c + 1;
            if (stop_at != 0x7FFFFFFF) atomicMin((int*)&L.ctr[CTR_KOUT], stop_at);
            if (W != 0.f && c < cap) {
                const float mx = sx / W, my = sy / W;
                float cxx = 0.f, cxy = 0.f, cyy = 0.f;
                for (int i = b0; i < e; ++i) {
                    const int p = (int)(L.key2[i] & 0xFFFFu);
                    const float w = L.w[p];
                    const float d0 = mx - L.mx[p];
                    const float d1 = my - L.my[p];
                    cxx += w * (L.xx[p] + d0 * d0);
                    cxy += w * (L.xy[p] + d0 * d1);
                    cyy += w * (L.yy[p] + d1 * d1);
                }
                out_slab[0 * cap + c] = W;
                out_slab[1 * cap + c] = mx;
                out_slab[2 * cap + c] = my;
                out_slab[3 * cap + c] = cxx / W;
                out_slab[4 * cap + c] = cxy / W;
                out_slab[5 * cap + c] = cyy / W;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int k = L.ctr[CTR_KOUT];
        if (k > n_clusters) k = n_clusters;
        L.ctr[CTR_KOUT] = k;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// the fused update + prune + merge kernel
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PHD_T) void phd_update_merge_kernel(UpdateArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const Lds L = lds_carve((lds_u8)lds_raw, A.S_cap, A.cap, A.MM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x;
    const DevConfig& cfg = A.cfg;
    const int cap = A.cap, S_cap = A.S_cap, M = A.M;
    const int src = A.parent[p];
    const int n_map = A.count_in[src];
    const float* __restrict__ in = A.map_in + (size_t)src * 6 * cap;
    float* __restrict__ out = A.map_out + (size_t)p * 6 * cap;
    const phd_pose pose = A.pose[p];

    if (tid < 16) L.ctr[tid] = 0;
    // measurements -> LDS (the reference keeps them in __constant__ Z[256], src/phdfilter.cu:120)
    for (int m = tid; m < M; m += PHD_T) {
        phd_measurement z = A.z[m];
        L.z_r[m] = z.range;
        L.z_b[m] = z.bearing;
        L.zok[m] = (z.label == 0 || !cfg.labeledMeasurements) ? 1u : 0u; // :1913
    }
    __syncthreads();

    // ---- classification + per-feature EKF terms -----------------------------------------------
    float pdw_local = 0.f; // sum_j pd_j w_j (cardinality_predict, :2160)
    {
        int n_in = 0, n_out0 = 0;
        for (int i0 = 0; i0 < n_map; i0 += PHD_T) {
            const int i = i0 + tid;
            int cls = -1;
            float w = 0, mx = 0, my = 0, pxx = 0, pxy = 0, pyy = 0;
            EkfTerms t;
            if (i < n_map) {
                w = in[0 * cap + i]; mx = in[1 * cap + i]; my = in[2 * cap + i];
                pxx = in[3 * cap + i]; pxy = in[4 * cap + i]; pyy = in[5 * cap + i];
                ekf_terms(mx, my, pxx, pxy, pyy, pose, cfg, t);
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
                L.f_r[j] = t.r; L.f_b[j] = t.b;
                L.f_s00[j] = t.s00; L.f_s12[j] = t.s12; L.f_s11[j] = t.s11;
                // log pd + log w + (-log 2pi - 0.5 log det)   (:1911,1916-1917), folded per feature
                const float lw0 = safe_log(t.pd) + safe_log(w);
                L.f_lwb[j] = lw0 - safe_log(6.2831855f) - 0.5f * safe_log(t.det);
                L.f_idx[j] = (u16)i;
                pdw_local += t.pd * w;
                // non-detection term (:2145-2148): prior with weight w(1-pd); prune test (:2314)
            } else if (cls == 0) {
                L.out_idx[off_out + __popcll(b_out & lanemask_lt())] = (u16)i;
            }
            // nearly-in-range features skip the update and join the merge (:3242-3257)
            {
                const bool keep = (cls == 2);
                const int slot = alloc_slots(keep, L.ctr);
                if (keep) store_survivor(L, slot, S_cap, w, mx, my, pxx, pxy, pyy, NEAR_U_BASE + i);
            }
            n_in += tot_in; n_out0 += tot_out;
            __syncthreads();
        }
        if (tid == 0) { L.ctr[CTR_NIN] = n_in; L.ctr[CTR_NOUT] = n_out0; }
    }
    __syncthreads();
    const int n_in = L.ctr[CTR_NIN];
    const int n_out0 = L.ctr[CTR_NOUT];

    // lane <-> measurement mapping
    int Mp = 1;
    while (Mp < M && Mp < 64) Mp <<= 1;          // lanes per feature group
    const int JS = 64 / Mp;                      // features in flight per wave
    const int m_tiles = (M + 63) / 64;
    const int lm = lane & (Mp - 1);
    const int js = lane / Mp;

    // ---- pass 1: normalisers ----------------------------------------------------------------------
    for (int mt = 0; mt < m_tiles; ++mt) {
        const int m = mt * 64 + lm;
        const bool mvalid = (m < M) && L.zok[m < M ? m : 0];
        const float zr = L.z_r[m < M ? m : 0], zb = L.z_b[m < M ? m : 0];
        float acc = 0.f;
        for (int jb = wave * JS; jb < n_in; jb += PHD_NW * JS) {
            const int j = jb + js;
            const int jj = j < n_in ? j : n_in - 1;
            const float i0 = zr - L.f_r[jj];
            const float i1 = wrap_angle(zb - L.f_b[jj]);
            const float dist = i0 * i0 * L.f_s00[jj] + i0 * i1 * L.f_s12[jj] + i1 * i1 * L.f_s11[jj]; // :1908-1910
            const float lw = L.f_lwb[jj] - 0.5f * dist;
            const float e = __expf(lw);                                                               // :2205
            acc += (mvalid && j < n_in) ? e : 0.f;
        }
        for (int off = Mp; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
        if (js == 0 && m < M) L.zpart[wave * A.MM + m] = acc;
    }
    __syncthreads();
    float lz_local = 0.f;
    for (int m = tid; m < M; m += PHD_T) {
        float sum = L.zpart[0 * A.MM + m];
#pragma unroll
        for (int wv = 1; wv < PHD_NW; ++wv) sum += L.zpart[wv * A.MM + m];
        sum += cfg.clutterDensity;                                                                    // :2213
        sum += cfg.birthWeight;                                                                       // :2214
        const float lz = safe_log(sum);                                                               // :2217
        L.logZ[m] = lz;
        lz_local += lz;                                                                               // :2251
    }
    {
        const float lz_sum = block_sum(lz_local, L.red, tid);
        const float pdw = block_sum(pdw_local, L.red, tid);
        // particle_weighting == 0 (:2260-2263): sum_m log Z_m - (sum_j pd_j w_j + M * birthWeight)
        if (tid == 0) A.dlogw[p] = lz_sum - (pdw + (float)M * cfg.birthWeight);
    }
    __syncthreads();

    // ---- pass 2: final weights; prune before store --------------------------------------------------
    // non-detection terms
    for (int j0 = 0; j0 < n_in; j0 += PHD_T) {
        const int j = j0 + tid;
        bool keep = false;
        float w = 0, mx = 0, my = 0, pxx = 0, pxy = 0, pyy = 0;
        if (j < n_in) {
            const int i = L.f_idx[j];
            w = in[0 * cap + i]; mx = in[1 * cap + i]; my = in[2 * cap + i];
            pxx = in[3 * cap + i]; pxy = in[4 * cap + i]; pyy = in[5 * cap + i];
            // pd of an in-range feature (:1849-1850)
            const float pd = (L.f_r[j] <= cfg.maxRange && fabsf(L.f_b[j]) <= cfg.maxBearing) ? cfg.pd : 0.f;
            w = w * (1 - pd);                                                                         // :2148
            keep = !(w < cfg.minFeatureWeight);                                                       // :2314
        }
        const int slot = alloc_slots(keep, L.ctr);
        if (keep) store_survivor(L, slot, S_cap, w, mx, my, pxx, pxy, pyy, j);
    }
    // detection terms
    for (int mt = 0; mt < m_tiles; ++mt) {
        const int m = mt * 64 + lm;
        const bool mvalid = (m < M);
        const bool zok = mvalid && L.zok[mvalid ? m : 0];
        const float zr = L.z_r[mvalid ? m : 0], zb = L.z_b[mvalid ? m : 0], lz = L.logZ[mvalid ? m : 0];
        for (int jb = wave * JS; jb < n_in; jb += PHD_NW * JS) {
            const int j = jb + js;
            const int jj = j < n_in ? j : n_in - 1;
            const float i0 = zr - L.f_r[jj];
            const float i1 = wrap_angle(zb - L.f_b[jj]);
            const float dist = i0 * i0 * L.f_s00[jj] + i0 * i1 * L.f_s12[jj] + i1 * i1 * L.f_s11[jj];
            const float lw = L.f_lwb[jj] - 0.5f * dist;
            const float w = zok ? __expf(lw - lz) : 0.f;                                              // :2242-2243
            const bool keep = mvalid && (j < n_in) && !(w < cfg.minFeatureWeight);                    // :2314
            const int slot = alloc_slots(keep, L.ctr);
            if (keep) {
                // rare: rebuild gain and Joseph covariance from the prior feature
                const int i = L.f_idx[jj];
                const float fmx = in[1 * cap + i], fmy = in[2 * cap + i];
                const float pxx = in[3 * cap + i], pxy = in[4 * cap + i], pyy = in[5 * cap + i];
                EkfTerms t;
                ekf_terms(fmx, fmy, pxx, pxy, pyy, pose, cfg, t);
                float oxx, oxy, oyy;
                joseph_cov(t, pxx, pxy, pyy, cfg, oxx, oxy, oyy);
                const float nmx = fmx + t.K0 * i0 + t.K2 * i1;                                       // :1903-1904
                const float nmy = fmy + t.K1 * i0 + t.K3 * i1;
                store_survivor(L, slot, S_cap, w, nmx, nmy, oxx, oxy, oyy, n_in + m * n_in + j);
            }
        }
    }
    // births (host loop src/phdfilter.cu:3470-3506; weight :2239-2243)
    for (int m0 = 0; m0 < M; m0 += PHD_T) {
        const int m = m0 + tid;
        bool keep = false;
        float w = 0, bmx = 0, bmy = 0, bxx = 0, bxy = 0, byy = 0;
        if (m < M) {
            w = L.zok[m] ? expf(safe_log(cfg.birthWeight) - L.logZ[m]) : 0.f;
            keep = !(w < cfg.minFeatureWeight);
            if (keep) {
                const float zr = L.z_r[m];
                const float theta = pose.ptheta + L.z_b[m];
                float sn, cs;
                sincosf(theta, &sn, &cs);
                const float dx = zr * cs, dy = zr * sn;
                bmx = pose.px + dx;
                bmy = pose.py + dy;
                const float J0 = dx / zr, J1 = dy / zr, J2 = -dy, J3 = dx;
                const float sr = cfg.stdRange * cfg.birthNoiseFactor, sb = cfg.stdBearing * cfg.birthNoiseFactor;
                const float vr = sr * sr, vb = sb * sb;
                bxx = J0 * J0 * vr + J2 * J2 * vb;
                bxy = J0 * J1 * vr + J2 * J3 * vb;
                byy = J1 * J1 * vr + J3 * J3 * vb;
            }
        }
        const int slot = alloc_slots(keep, L.ctr);
        if (keep) store_survivor(L, slot, S_cap, w, bmx, bmy, bxx, bxy, byy, n_in + M * n_in + m);
    }
    __syncthreads();
    int n_surv = L.ctr[CTR_NSURV];
    unsigned status = 0;
    if (n_surv > S_cap) { n_surv = S_cap; status |= PHD_STATUS_SURVIVOR_OVERFLOW; }
    if (L.ctr[CTR_OVERFLOW]) status |= PHD_STATUS_SURVIVOR_OVERFLOW;

    // optional inspection copy of the survivors (parity tests)
    if (A.dbg_surv) {
        float* d = A.dbg_surv + (size_t)p * 6 * S_cap;
        int* du = A.dbg_u + (size_t)p * S_cap;
        for (int i = tid; i < n_surv; i += PHD_T) {
            d[0 * S_cap + i] = L.w[i]; d[1 * S_cap + i] = L.mx[i]; d[2 * S_cap + i] = L.my[i];
            d[3 * S_cap + i] = L.xx[i]; d[4 * S_cap + i] = L.xy[i]; d[5 * S_cap + i] = L.yy[i];
            du[i] = L.u[i];
        }
        if (tid == 0) { A.dbg_n[p] = n_surv; A.dbg_nin[p] = n_in; }
    }
    __syncthreads();

    // ---- merge ----------------------------------------------------------------------------------------
    if (cfg.distanceMetric == 0) merge_in_lds<false>(L, S_cap, n_surv, cfg, out, cap, tid);
    else merge_in_lds<true>(L, S_cap, n_surv, cfg, out, cap, tid);
    int k_out = L.ctr[CTR_KOUT];
    if (k_out > cap) { k_out = cap; status |= PHD_STATUS_MAP_OVERFLOW; }
    // append the untouched out-of-range features (src/phdfilter.cu:3311-3318)
    int n_app = n_out0;
    if (k_out + n_app > cap) { n_app = cap - k_out; status |= PHD_STATUS_MAP_OVERFLOW; }
    for (int i = tid; i < n_app; i += PHD_T) {
        const int s = L.out_idx[i];
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) out[pl * cap + k_out + i] = in[pl * cap + s];
    }
    if (tid == 0) {
        if (A.parent_reset) A.parent_reset[p] = p; // the output slab of particle p is its own again
        A.count_out[p] = k_out + n_app;
        if (status) atomicOr(A.status, status);
        atomicMax(A.max_surv, L.ctr[CTR_NSURV]);
        atomicMax(A.max_map, k_out + n_out0);
    }
}

// ------------------------------------------------------------------------------------------
// vehicle predict (phdPredictKernelAckerman, src/phdfilter.cu:785-825)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void phd_predict_kernel(const phd_pose* __restrict__ in, phd_pose* __restrict__ out, int n,
                                   phd_ackerman_control u, const phd_ackerman_noise* __restrict__ noise,
                                   u64 seed, u64 counter, DevConfig cfg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    phd_pose o = in[i];
    float n_alpha, n_encoder;
    if (noise) {
        n_alpha = noise[i].n_alpha;
        n_encoder = noise[i].n_encoder;
    } else {
        // counter-based generator: Box-Muller on two splitmix64 outputs (replaces rng.cpp's
        // wall-clock seeded boost::mt19937; draw order (n_alpha, n_encoder), phdfilter.cu:1148-1152)
        u64 a = splitmix64(seed ^ splitmix64(counter * 0x100000001B3ull + (u64)i * 2ull));
        u64 b = splitmix64(a);
        float u1 = ((float)((a >> 40) + 1)) * (1.0f / 16777216.0f);
        float u2 = ((float)(b >> 40)) * (1.0f / 16777216.0f);
        float rad = sqrtf(-2.f * logf(u1));
        float sn, cs;
        sincosf(6.2831855f * u2, &sn, &cs);
        n_alpha = cfg.stdAlpha * (rad * cs);
        n_encoder = cfg.stdEncoder * (rad * sn);
    }
    const float ve = u.v_encoder + n_encoder;                                   // :802
    const float al = u.alpha + n_alpha;                                         // :803
    const float tn = tanf(al);
    const float vc = ve / (1 - tn * cfg.h / cfg.l);                             // :804
    float sn, cs;
    sincosf(o.ptheta, &sn, &cs);
    const float xc_dot = vc * cs, yc_dot = vc * sn;                             // :805-806
    const float thetac_dot = vc * tn / cfg.l;                                   // :807
    const float dt = cfg.dt / cfg.subdividePredict;                             // :808
    phd_pose nw;
    nw.px = o.px + dt * (xc_dot - thetac_dot * (cfg.a * sn + cfg.b * cs));     // :809-812
    nw.py = o.py + dt * (yc_dot + thetac_dot * (cfg.a * cs - cfg.b * sn));     // :813-816
    nw.ptheta = wrap_angle(o.ptheta + dt * thetac_dot);                         // :817
    nw.vx = 0; nw.vy = 0; nw.vtheta = 0;                                        // :818-820
    out[i] = nw;
}

// ------------------------------------------------------------------------------------------
// particle weights: accumulate, logSumExp normalise, nEff, resample (one workgroup)
// ------------------------------------------------------------------------------------------
// portable exp for the resampling CDF: IEEE basic operations only (mul, fma, rint, ldexp), so
// the double it returns is the same on every conforming CPU and GPU (see oracle/scphd_cpu.c).
__device__ __forceinline__ double det_exp(float xf)
{
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double x = (double)xf;
    if (!(x >= -700.0)) return (x != x) ? x : 0.0;
    if (x > 700.0) return (double)INFINITY;
    double kd = rint(x * LOG2E);
    double r = fma(-kd, LN2_HI, x);
    r = fma(-kd, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}

#define PHD_WT 1024
#define PHD_CDF_CHUNK 2048

__device__ __forceinline__ float block_reduce_w(float v, float* sc, int tid, bool is_max)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float o = __shfl_xor(v, off);
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

__global__ __launch_bounds__(PHD_WT) void phd_weights_kernel(WeightArgs A)
{
    __shared__ float sc[PHD_WT / 64];
    __shared__ int s_flag;
    __shared__ int s_argmax;
    __shared__ double s_chunk[PHD_CDF_CHUNK];
    __shared__ double s_carry;
    __shared__ double s_bestv[PHD_WT / 64];
    __shared__ int s_besti[PHD_WT / 64];
    const int tid = threadIdx.x;
    const int n = A.n;          // weights in the vector being normalised (global count for multi-GPU)
    float* logw = A.logw;       // [n] working / output vector (== logw_in unless the filter is frozen)
    // 1. accumulate the increments of the last update (src/phdfilter.cu:3741-3744)
    if ((A.mode & W_ACCUMULATE) || A.logw_in != logw) {
        for (int i = tid; i < n; i += PHD_WT) {
            float w = A.logw_in[i];
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
        mx = block_reduce_w(mx, sc, tid, true);
        float s = 0.f;
        for (int i = tid; i < n; i += PHD_WT) s += expf(logw[i] - mx);
        s = block_reduce_w(s, sc, tid, false);
        const float lse = safe_log(s) + mx;
        for (int i = tid; i < n; i += PHD_WT) logw[i] -= lse;
        __syncthreads();
    }
    // 3. nEff = 1 / sum exp(2w) / N (src/main.cpp:1281-1284)
    float s2 = 0.f;
    for (int i = tid; i < n; i += PHD_WT) s2 += expf(2 * logw[i]);
    s2 = block_reduce_w(s2, sc, tid, false);
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
        for (int j = tid; j < n_new; j += PHD_WT) {
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
    //    CDF: p_i = det_exp(w_i), accumulated SEQUENTIALLY in double in index order (:463,495) by one
    //    lane — parallel scans round differently and would break bit-exactness across ranks/CPU.
    double* cdf = A.cdf;   // [n] global scratch
    const double interval = 1.0 / n_new;
    double best = -1.0;
    int besti = 0x7FFFFFFF;
    for (int c0 = 0; c0 < n; c0 += PHD_CDF_CHUNK) {
        const int m = (n - c0 < PHD_CDF_CHUNK) ? (n - c0) : PHD_CDF_CHUNK;
        for (int i = tid; i < m; i += PHD_WT) {
            const double e = det_exp(logw[c0 + i]);
            s_chunk[i] = e;
            if (e > best) { best = e; besti = c0 + i; } // strided ascending: keeps the lowest index per lane
        }
        __syncthreads();
        if (tid == 0) {
            double c = (c0 == 0) ? 0.0 : s_carry;
            int i = 0;
            for (; i + 8 <= m; i += 8) {
                const double e0 = s_chunk[i], e1 = s_chunk[i + 1], e2 = s_chunk[i + 2], e3 = s_chunk[i + 3];
                const double e4 = s_chunk[i + 4], e5 = s_chunk[i + 5], e6 = s_chunk[i + 6], e7 = s_chunk[i + 7];
                c += e0; s_chunk[i] = c;
                c += e1; s_chunk[i + 1] = c;
                c += e2; s_chunk[i + 2] = c;
                c += e3; s_chunk[i + 3] = c;
                c += e4; s_chunk[i + 4] = c;
                c += e5; s_chunk[i + 5] = c;
                c += e6; s_chunk[i + 6] = c;
                c += e7; s_chunk[i + 7] = c;
            }
            for (; i < m; ++i) { c += s_chunk[i]; s_chunk[i] = c; }
            s_carry = c;
        }
        __syncthreads();
        if (n > PHD_CDF_CHUNK) {
            for (int i = tid; i < m; i += PHD_WT) cdf[c0 + i] = s_chunk[i];
            __syncthreads();
        }
    }
    // arg-max of p (first maximum, strict '>'), used by the overflow guard (:475-494)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(besti, off);
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
    const bool in_lds = (n <= PHD_CDF_CHUNK);
    const double ctot = s_carry;
    for (int j = tid; j < n_new; j += PHD_WT) {
        const double u = (A.n_uniforms == 1) ? A.u0 : A.uniforms[j];
        const double r = j * interval + u * interval;                                                  // :468
        int idx;
        if (r > ctot) {
            idx = s_argmax;                                                                            // :475-494
        } else {
            // smallest i with !(r > cdf[i])  ==  where the reference's "while (r > c) i++" stops
            int lo = 0, hi = n - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const double cm = in_lds ? s_chunk[mid] : cdf[mid];
                if (r > cm) lo = mid + 1; else hi = mid;
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
        const float r = block_reduce_w(acc[k], sc, tid, false);
        if (tid == 0) pose_out[k] = r;
    }
    // arg-max with ties to the lowest index (strict '>' scan in the reference)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
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
__global__ void phd_export_kernel(const float* __restrict__ slabs, const int* __restrict__ counts,
                                  const int* __restrict__ parent, const phd_pose* __restrict__ poses,
                                  const int* __restrict__ which, unsigned char* __restrict__ buf, int cap, size_t stride)
{
    const int k = blockIdx.x;
    const int p = which[k];
    const int src = parent ? parent[p] : p;
    float* o = (float*)(buf + (size_t)k * stride);
    const int n = counts[src];
    if (threadIdx.x < 6) o[threadIdx.x] = ((const float*)&poses[p])[threadIdx.x];
    if (threadIdx.x == 6) ((int*)o)[6] = n;
    const float* s = slabs + (size_t)src * 6 * cap;
    for (int i = threadIdx.x; i < 6 * cap; i += blockDim.x) o[8 + i] = ((i % cap) < n) ? s[i] : 0.f;
}

__global__ void phd_import_kernel(float* __restrict__ slabs, int* __restrict__ counts, phd_pose* __restrict__ poses,
                                  const int* __restrict__ which, const unsigned char* __restrict__ buf, int cap,
                                  size_t stride)
{
    const int k = blockIdx.x;
    const int p = which[k];
    const float* o = (const float*)(buf + (size_t)k * stride);
    if (threadIdx.x < 6) ((float*)&poses[p])[threadIdx.x] = o[threadIdx.x];
    if (threadIdx.x == 6) counts[p] = ((const int*)o)[6];
    float* s = slabs + (size_t)p * 6 * cap;
    for (int i = threadIdx.x; i < 6 * cap; i += blockDim.x) s[i] = o[8 + i];
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

__global__ void phd_iota_kernel(int* a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}

// ------------------------------------------------------------------------------------------
// launchers (called from phd_api.cpp; plain C++ signatures, no <<<>>> outside this file)
// ------------------------------------------------------------------------------------------
hipError_t launch_update_merge(const UpdateArgs& a, int n_particles, size_t lds_bytes, hipStream_t st)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)phd_update_merge_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(phd_update_merge_kernel, dim3(n_particles), dim3(PHD_T), lds_bytes, st, a);
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

hipError_t launch_weights(const WeightArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(phd_weights_kernel, dim3(1), dim3(PHD_WT), 0, st, a);
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

hipError_t launch_state(const phd_pose* poses, const float* logw, int n, float* pose_out, int* argmax_out,
                        hipStream_t st)
{
    hipLaunchKernelGGL(phd_state_kernel, dim3(1), dim3(PHD_WT), 0, st, poses, logw, n, pose_out, argmax_out);
    return hipGetLastError();
}

hipError_t launch_export(const float* slabs, const int* counts, const int* parent, const phd_pose* poses,
                         const int* which, void* buf, int cap, size_t stride, int n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_export_kernel, dim3(n), dim3(256), 0, st, slabs, counts, parent, poses, which,
                       (unsigned char*)buf, cap, stride);
    return hipGetLastError();
}

hipError_t launch_import(float* slabs, int* counts, phd_pose* poses, const int* which, const void* buf, int cap,
                         size_t stride, int n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(phd_import_kernel, dim3(n), dim3(256), 0, st, slabs, counts, poses, which,
                       (const unsigned char*)buf, cap, stride);
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

hipError_t launch_iota(int* a, int n, hipStream_t st)
{
    hipLaunchKernelGGL(phd_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
    return hipGetLastError();
}

} // namespace phd
