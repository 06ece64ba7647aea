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
#include "phd_detexp.h"

namespace phd {

#ifndef PHD_NW
#define PHD_NW 8            // waves per workgroup (512 threads: two waves per SIMD hide LDS/ALU latency
                            // when a CU holds a single particle; throughput-neutral at 4096 particles)
#endif
#define PHD_T (64 * PHD_NW)
#ifndef PHD_MIN_WAVES
#define PHD_MIN_WAVES 4      // launch bound: waves per SIMD the register allocation must allow
#endif
#define PHD_COLS (64 / PHD_NW) // window columns (= candidate seeds) owned by one wave
#define PHD_SMALL_S 256        // survivor counts up to this take the single-shot merge (merge_small)
#define NEAR_U_BASE 0x40000000
// phase stamps of the diagnostic instantiation (100 MHz s_memrealtime), thread 0 of each workgroup
#define STAMP(k) do { if (STAMPS && tid == 0) st[k] = __builtin_amdgcn_s_memrealtime(); } while (0)

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
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

// ------------------------------------------------------------------------------------------
// cross-lane exchange on the VALU.  hipcc lowers __shfl_xor / __shfl_up to ds_bpermute_b32 (an LDS-pipe
// round trip of ~100 cycles plus the address VGPR); for the partners the sorts, scans and reductions
// need, gfx950 has single-issue VALU forms (encodings verified on hardware by tools/dpp_probe.hip):
//   lane ^ 1, ^ 2   DPP quad_perm            lane ^ 16   v_permlane16_swap + select
//   lane ^ 4        two DPP row_ror (banks)  lane ^ 32   v_permlane32_swap + select
//   lane ^ 8        DPP row_ror:8            lane - 1    DPP wave_shr:1
// Same values as the shuffles they replace, bit for bit.
// ------------------------------------------------------------------------------------------
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));

template <int CTRL, int BANKS>
__device__ __forceinline__ u32 dpp_mov(u32 old, u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANKS, false);
}

template <int OFF>
__device__ __forceinline__ u32 xor_lane_c(u32 v)
{
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "power of two < 64");
    if (OFF == 1) return dpp_mov<0xB1, 0xF>(v, v);                 // quad_perm [1,0,3,2]
    if (OFF == 2) return dpp_mov<0x4E, 0xF>(v, v);                 // quad_perm [2,3,0,1]
    if (OFF == 4) return dpp_mov<0x124, 0xA>(dpp_mov<0x12C, 0x5>(v, v), v); // row_ror:12 into banks 0,2; row_ror:4 into 1,3
    if (OFF == 8) return dpp_mov<0x128, 0xF>(v, v);                // row_ror:8
    if (OFF == 16) {
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (__lane_id() & 16) ? r.x : r.y;
    }
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (__lane_id() & 32) ? r.x : r.y;
}

// off: wave-uniform power of two in 1..32 (a scalar branch selects the encoding)
__device__ __forceinline__ u32 xor_lane(u32 v, int off)
{
    switch (off) {
    case 1: return xor_lane_c<1>(v);
    case 2: return xor_lane_c<2>(v);
    case 4: return xor_lane_c<4>(v);
    case 8: return xor_lane_c<8>(v);
    case 16: return xor_lane_c<16>(v);
    default: return xor_lane_c<32>(v);
    }
}
__device__ __forceinline__ float xor_lane(float v, int off) { return __uint_as_float(xor_lane(__float_as_uint(v), off)); }
__device__ __forceinline__ int xor_lane(int v, int off) { return (int)xor_lane((u32)v, off); }
__device__ __forceinline__ double xor_lane(double v, int off)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u64 r = ((u64)xor_lane((u32)(b >> 32), off) << 32) | xor_lane((u32)b, off);
    return __longlong_as_double((long long)r);
}
template <int OFF> __device__ __forceinline__ float xor_lane_c(float v) { return __uint_as_float(xor_lane_c<OFF>(__float_as_uint(v))); }

// value of lane l (wave-uniform l) as a scalar broadcast
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// value of lane + 1 (lane 63 keeps its own), = __shfl_down(v, 1)
__device__ __forceinline__ u32 lane_down1(u32 v) { return dpp_mov<0x130, 0xF>(v, v); }
__device__ __forceinline__ float lane_down1(float v) { return __uint_as_float(lane_down1(__float_as_uint(v))); }
__device__ __forceinline__ int lane_down1(int v) { return (int)lane_down1((u32)v); }
// value of lane - 1 (lane 0 keeps its own), = __shfl_up(v, 1)
__device__ __forceinline__ u32 lane_up1(u32 v) { return dpp_mov<0x138, 0xF>(v, v); }
__device__ __forceinline__ float lane_up1(float v) { return __uint_as_float(lane_up1(__float_as_uint(v))); }
__device__ __forceinline__ int lane_up1(int v) { return (int)lane_up1((u32)v); }

// inclusive prefix sum over the wave (integers: any association is exact): Kogge-Stone inside each row of
// 16 lanes with DPP row_shr (zero fill), then the row totals through row_bcast15 / row_bcast31
template <int CTRL, int ROWS>
__device__ __forceinline__ u32 dpp_zero(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, true);
}
__device__ __forceinline__ u32 wave_incl_scan(u32 v)
{
    v += dpp_zero<0x111, 0xF>(v); v += dpp_zero<0x112, 0xF>(v);
    v += dpp_zero<0x114, 0xF>(v); v += dpp_zero<0x118, 0xF>(v);
    v += dpp_zero<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    v += dpp_zero<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ u64 wave_incl_scan(u64 v)
{
#define PHD_SCAN64_STEP(CTRL, ROWS)                                                                      \
    v += ((u64)dpp_zero<CTRL, ROWS>((u32)(v >> 32)) << 32) | dpp_zero<CTRL, ROWS>((u32)v);
    PHD_SCAN64_STEP(0x111, 0xF) PHD_SCAN64_STEP(0x112, 0xF) PHD_SCAN64_STEP(0x114, 0xF) PHD_SCAN64_STEP(0x118, 0xF)
    PHD_SCAN64_STEP(0x142, 0xA) PHD_SCAN64_STEP(0x143, 0xC)
#undef PHD_SCAN64_STEP
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += xor_lane_c<32>(v); v += xor_lane_c<16>(v); v += xor_lane_c<8>(v);
    v += xor_lane_c<4>(v); v += xor_lane_c<2>(v); v += xor_lane_c<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, xor_lane_c<32>(v)); v = fmaxf(v, xor_lane_c<16>(v)); v = fmaxf(v, xor_lane_c<8>(v));
    v = fmaxf(v, xor_lane_c<4>(v)); v = fmaxf(v, xor_lane_c<2>(v)); v = fmaxf(v, xor_lane_c<1>(v));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, (int)xor_lane_c<32>((u32)v)); v = max(v, (int)xor_lane_c<16>((u32)v)); v = max(v, (int)xor_lane_c<8>((u32)v));
    v = max(v, (int)xor_lane_c<4>((u32)v)); v = max(v, (int)xor_lane_c<2>((u32)v)); v = max(v, (int)xor_lane_c<1>((u32)v));
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

// two block sums behind one pair of barriers (each value takes exactly block_sum's tree: same bits);
// scratch: 2 * PHD_NW floats
__device__ __forceinline__ void block_sum2(float a, float b, LDS_T(float)* scratch, int tid, float& ra, float& rb)
{
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((tid & 63) == 0) { scratch[tid >> 6] = a; scratch[PHD_NW + (tid >> 6)] = b; }
    __syncthreads();
    ra = scratch[0]; rb = scratch[PHD_NW];
#pragma unroll
    for (int w = 1; w < PHD_NW; ++w) { ra += scratch[w]; rb += scratch[PHD_NW + w]; }
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
    u32 out_idx, z_r, z_b, logZ, zpart, zok, bgeo, part, win, red, ctr;
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
    const u32 feat = align16u(16u * (u32)C) + align16u(8u * (u32)C) + align16u(2u * (u32)C) + 2u * align16u(16u * (u32)C) +
                     align16u(4u * (u32)C);
    const u32 sort1 = 3u * sv;
    const u32 sort2 = sv + align16u(4u * (u32)(S + 1));
    u32 amax = feat > sort1 ? feat : sort1;
    amax = amax > sort2 ? amax : sort2;
    const u32 small = 2u * PHD_SMALL_S * 32u + 64u + 2u * PHD_SMALL_S * 16u; // merge_small(): rows, member columns, seeds, staged planes
    amax = amax > small ? amax : small;
    p += amax;
    o.out_idx = p; p += align16u(2u * (u32)C);
    o.z_r = p; p += align16u(4u * (u32)MM);
    o.z_b = p; p += align16u(4u * (u32)MM);
    o.logZ = p; p += align16u(4u * (u32)MM);
    o.zpart = p; p += align16u(4u * PHD_NW * (u32)MM);
    o.zok = p; p += align16u(4u * (u32)MM);
    o.bgeo = p; p += align16u(20u * (u32)MM);
    o.part = p; p += 4u * PHD_NW * 64u;
    o.win = p; p += 4u * 7u * 64u;
    o.red = p; p += align16u(4u * (2 * PHD_NW + 4));
    o.ctr = p; p += 4u * 32u;
    o.total = p;
    return o;
}

struct Lds {
    // survivors (SoA), S entries each
    lds_f32 w, mx, my, xx, xy, yy;
    lds_f32 tr; // trace of the covariance (+inf if not SPD): cheap far-pair filter of the merge
    lds_i32 u; // slab index (sort tie-break); after the sort: cluster assignment
    // aliased region
    LDS_T(v4f)* f_a;                           // per in-range feature: (r, b, S00, S01+S10)
    LDS_T(v2f)* f_c;                           //                       (S11, folded log-weight base)
    lds_u16 f_idx;                                // map index of in-range feature j
    LDS_T(v4f)* f_k;                              // Kalman gain K0..K3 of in-range feature j
    LDS_T(v4f)* f_p;                              // Joseph-form updated covariance (the same for every measurement) and prior mean x: (xx, xy, yy, mx)
    lds_f32 f_my;                                 // prior mean y
    lds_u32 khi, klo, pay;                        // sort 1
    lds_u32 key2;                                 // sort 2
    lds_i32 seg;                                  // cluster starts, S+1
    LDS_T(u64)* srow;                             // merge_small: [256][4] closeness to earlier positions
    LDS_T(u64)* scol;                             // merge_small: [256][4] members of the cluster seeded at a position
    LDS_T(u64)* sseed;                            // merge_small: [4] seed mask
    LDS_T(v4f)* sA;                               // merge_small: [256] (mx, my, 0.505 T tr, w) in sorted order
    LDS_T(v4f)* sB;                               // merge_small: [256] (xx, xy, yy, -)
    // not aliased
    lds_u16 out_idx;                  // C
    lds_f32 z_r, z_b, logZ, zpart;    // MM, MM, MM, 4*MM
    lds_u32 zok;                      // MM
    lds_f32 bgeo;                     // 5*MM: birth mean and covariance per measurement
    lds_u32 part;                     // 4*64 row parts of the window closeness matrix
    lds_f32 win;                      // 7*64: the window's candidates (pos, mx, my, tr, xx, xy, yy)
    lds_f32 red;                      // PHD_NW + 4
    lds_i32 ctr;                      // 32 counters
};

__device__ __forceinline__ Lds lds_carve(lds_u8 base, int S, int C, int MM)
{
    const LdsOffsets o = lds_offsets(S, C, MM);
    Lds L;
    L.w = (lds_f32)(base + o.w); L.mx = (lds_f32)(base + o.mx); L.my = (lds_f32)(base + o.my);
    L.xx = (lds_f32)(base + o.xx); L.xy = (lds_f32)(base + o.xy); L.yy = (lds_f32)(base + o.yy);
    L.tr = (lds_f32)(base + o.tr); L.u = (lds_i32)(base + o.u);
    u32 f = o.alias;
    L.f_a = (LDS_T(v4f)*)(base + f); f += align16u(16u * (u32)C);
    L.f_c = (LDS_T(v2f)*)(base + f); f += align16u(8u * (u32)C);
    L.f_idx = (lds_u16)(base + f); f += align16u(2u * (u32)C);
    L.f_k = (LDS_T(v4f)*)(base + f); f += align16u(16u * (u32)C);
    L.f_p = (LDS_T(v4f)*)(base + f); f += align16u(16u * (u32)C);
    L.f_my = (lds_f32)(base + f);
    const u32 sv = align16u(4u * (u32)S);
    L.khi = (lds_u32)(base + o.alias);
    L.klo = (lds_u32)(base + o.alias + sv);
    L.pay = (lds_u32)(base + o.alias + 2u * sv);
    L.key2 = (lds_u32)(base + o.alias);
    L.seg = (lds_i32)(base + o.alias + sv);
    L.srow = (LDS_T(u64)*)(base + o.alias);
    L.scol = (LDS_T(u64)*)(base + o.alias + PHD_SMALL_S * 32u);
    L.sseed = (LDS_T(u64)*)(base + o.alias + 2u * PHD_SMALL_S * 32u);
    L.sA = (LDS_T(v4f)*)(base + o.alias + 2u * PHD_SMALL_S * 32u + 64u);
    L.sB = (LDS_T(v4f)*)(base + o.alias + 2u * PHD_SMALL_S * 32u + 64u + PHD_SMALL_S * 16u);
    L.out_idx = (lds_u16)(base + o.out_idx);
    L.z_r = (lds_f32)(base + o.z_r); L.z_b = (lds_f32)(base + o.z_b); L.logZ = (lds_f32)(base + o.logZ);
    L.zpart = (lds_f32)(base + o.zpart); L.zok = (lds_u32)(base + o.zok);
    L.bgeo = (lds_f32)(base + o.bgeo);
    L.part = (lds_u32)(base + o.part);
    L.win = (lds_f32)(base + o.win);
    L.red = (lds_f32)(base + o.red);
    L.ctr = (lds_i32)(base + o.ctr);
    return L;
}

size_t update_lds_bytes(int S, int C, int MM) { return lds_offsets(S, C, MM).total; }
int update_fuse_max_particles() { return PHD_T * 2; } // weights_body<PHD_T, 2> of the fused step

enum { CTR_NSURV = 0, CTR_NIN = 1, CTR_NOUT = 2, CTR_OVERFLOW = 3, CTR_KOUT = 4, CTR_NHEAD = 5, CTR_TMP = 6 /* ..+PHD_NW*2 <= 22 */,
       CTR_NPAIR = 24 /* merge_small: candidate pairs listed for the exact closeness test */, CTR_NNEAR = 30 };
#define PHD_CAND_SEG (1920 / PHD_NW) // u16 entries per wave in the (part, win) arrays, free until the merge

// append one survivor; slot allocation is wave-aggregated (one LDS atomic per wave per call site)
__device__ __forceinline__ int alloc_slots(bool keep, lds_i32 ctr)
{
    u64 bal = __ballot(keep);
    int slot = -1;
    if (bal) {
        int base = 0;
        if ((u64)(1ull << __lane_id()) == (bal & (~bal + 1))) base = atomicAdd((int*)&ctr[CTR_NSURV], __popcll(bal));
        int leader = __builtin_ctzll(bal);
        base = __builtin_amdgcn_readlane(base, leader);
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
// bitonic sorts held in registers: thread t owns elements i = t*E + e (e < E), n = sorted prefix
// (power of two, <= 256*E).  Strides below E are compare-exchanges between a thread's own
// registers, strides below 64*E are wave shuffles (no LDS traffic, no barrier); only the few
// strides that cross waves go through LDS.
// ------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void reg_sort_desc64(u32 (&khi)[E], u32 (&klo)[E], u32 (&pay)[E], int n, int tid,
                                                lds_u32 xhi, lds_u32 xlo, lds_u32 xpay)
{
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E * 64) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    if (i < n) { xhi[i] = khi[e]; xlo[i] = klo[e]; xpay[i] = pay[e]; }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    if (i < n) {
                        const int l = i ^ j;
                        const u64 mine = ((u64)khi[e] << 32) | klo[e];
                        const u64 oth = ((u64)xhi[l] << 32) | xlo[l];
                        const bool want_max = (((i & k) == 0) == ((i & j) == 0));
                        if (want_max ? (oth > mine) : (oth < mine)) { khi[e] = (u32)(oth >> 32); klo[e] = (u32)oth; pay[e] = xpay[l]; }
                    }
                }
                __syncthreads();
            } else if (j >= E) {
                const int lm = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    // (ds_bpermute on purpose: this sort runs when the kernel is VALU-bound, the LDS pipe is idle)
                    const u32 ohi = __shfl_xor(khi[e], lm), olo = __shfl_xor(klo[e], lm), op = __shfl_xor(pay[e], lm);
                    const u64 mine = ((u64)khi[e] << 32) | klo[e];
                    const u64 oth = ((u64)ohi << 32) | olo;
                    const bool want_max = (((i & k) == 0) == ((i & j) == 0));
                    if (want_max ? (oth > mine) : (oth < mine)) { khi[e] = ohi; klo[e] = olo; pay[e] = op; }
                }
            } else {
#pragma unroll
                for (int jj = E / 2; jj > 0; jj >>= 1) {
                    if (j == jj) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            if ((e & jj) == 0) {
                                const int i = tid * E + e;
                                const u64 a = ((u64)khi[e] << 32) | klo[e];
                                const u64 b = ((u64)khi[e | jj] << 32) | klo[e | jj];
                                const bool desc = ((i & k) == 0);
                                if (desc ? (a < b) : (a > b)) {
                                    const u32 th = khi[e], tl = klo[e], tp = pay[e];
                                    khi[e] = khi[e | jj]; klo[e] = klo[e | jj]; pay[e] = pay[e | jj];
                                    khi[e | jj] = th; klo[e | jj] = tl; pay[e | jj] = tp;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}


// keys-only variant of reg_sort_desc64 (the payload rides in the low 16 bits of the key)
template <int E>
__device__ __forceinline__ void reg_sort_desc64k(u32 (&khi)[E], u32 (&klo)[E], int n, int tid, lds_u32 xhi, lds_u32 xlo)
{
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E * 64) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    if (i < n) { xhi[i] = khi[e]; xlo[i] = klo[e]; }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    if (i < n) {
                        const int l = i ^ j;
                        const u64 mine = ((u64)khi[e] << 32) | klo[e];
                        const u64 oth = ((u64)xhi[l] << 32) | xlo[l];
                        const bool want_max = (((i & k) == 0) == ((i & j) == 0));
                        const u64 r = want_max ? (oth > mine ? oth : mine) : (oth < mine ? oth : mine);
                        khi[e] = (u32)(r >> 32); klo[e] = (u32)r;
                    }
                }
                __syncthreads();
            } else if (j >= E) {
                const int lm = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    const u32 ohi = __shfl_xor(khi[e], lm), olo = __shfl_xor(klo[e], lm);
                    const u64 mine = ((u64)khi[e] << 32) | klo[e];
                    const u64 oth = ((u64)ohi << 32) | olo;
                    const bool want_max = (((i & k) == 0) == ((i & j) == 0));
                    const u64 r = want_max ? (oth > mine ? oth : mine) : (oth < mine ? oth : mine);
                    khi[e] = (u32)(r >> 32); klo[e] = (u32)r;
                }
            } else {
#pragma unroll
                for (int jj = E / 2; jj > 0; jj >>= 1) {
                    if (j == jj) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            if ((e & jj) == 0) {
                                const int i = tid * E + e;
                                const u64 a = ((u64)khi[e] << 32) | klo[e];
                                const u64 b = ((u64)khi[e | jj] << 32) | klo[e | jj];
                                const bool desc = ((i & k) == 0);
                                const u64 hi = a > b ? a : b, lo = a > b ? b : a;
                                const u64 first = desc ? hi : lo, second = desc ? lo : hi;
                                khi[e] = (u32)(first >> 32); klo[e] = (u32)first;
                                khi[e | jj] = (u32)(second >> 32); klo[e | jj] = (u32)second;
                            }
                        }
                    }
                }
            }
        }
    }
}

// sort 1 of the merge: survivors by (weight desc, slab index asc), then permute the SoA arrays into
// that order (in place, staged through registers) and clear the assignment array.  The slab index of a
// nearly-in-range feature is n_update + its map index.  When every slab index fits 16 bits the key is
// (weight | ~slab index | slot) in 64 bits and nothing but the key is sorted.
template <int E>
__device__ __forceinline__ void sort_survivors(const Lds& L, int S, int n_pad, int tid, int n_update, bool packed)
{
    u32 khi[E], klo[E], pay[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < S) {
            const int u0 = L.u[i];
            const u32 u = (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0);
            khi[e] = orderable(L.w[i]);
            klo[e] = packed ? (((0xFFFFu - u) << 16) | (u32)i) : (0xFFFFFFFFu - u);
            pay[e] = (u32)i;
        } else { khi[e] = 0; klo[e] = 0; pay[e] = 0; }
    }
    if (packed) {
        reg_sort_desc64k<E>(khi, klo, n_pad, tid, L.khi, L.klo);
#pragma unroll
        for (int e = 0; e < E; ++e) pay[e] = klo[e] & 0xFFFFu;
    } else {
        reg_sort_desc64<E>(khi, klo, pay, n_pad, tid, L.khi, L.klo, L.pay);
    }
    float rw[E], rmx[E], rmy[E], rxx[E], rxy[E], ryy[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < S) {
            const int s = (int)pay[e];
            rw[e] = L.w[s]; rmx[e] = L.mx[s]; rmy[e] = L.my[s];
            rxx[e] = L.xx[s]; rxy[e] = L.xy[s]; ryy[e] = L.yy[s];
        }
    }
    __syncthreads(); // every read of the old order precedes every write of the new one
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < S) {
            L.w[i] = rw[e]; L.mx[i] = rmx[e]; L.my[i] = rmy[e];
            L.xx[i] = rxx[e]; L.xy[i] = rxy[e]; L.yy[i] = ryy[e];
            const bool spd = (rxx[e] > 0.f) && (ryy[e] > 0.f) && (rxx[e] * ryy[e] - rxy[e] * rxy[e] > 0.f);
            L.tr[i] = spd ? (rxx[e] + ryy[e]) : INFINITY;
            L.u[i] = -1; // unassigned
        }
    }
    __syncthreads();
}

// small mixtures (S <= workgroup size): rank by counting instead of a sorting network — every thread
// reads all keys (LDS broadcast reads, independent, no barriers inside); a survivor's rank is the
// number of keys that sort before its own
__device__ __forceinline__ void rank_sort_survivors(const Lds& L, int S, int tid, int n_update)
{
    u32 mh = 0, ml = 0;
    float rw = 0, rmx = 0, rmy = 0, rxx = 0, rxy = 0, ryy = 0;
    if (tid < S) {
        const int u0 = L.u[tid];
        mh = orderable(L.w[tid]);
        ml = 0xFFFFFFFFu - (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0);
        L.khi[tid] = mh; L.klo[tid] = ml;
        rw = L.w[tid]; rmx = L.mx[tid]; rmy = L.my[tid]; rxx = L.xx[tid]; rxy = L.xy[tid]; ryy = L.yy[tid];
    }
    __syncthreads();
    if (tid < S) {
        const u64 mine = ((u64)mh << 32) | ml;
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < S; ++j) {
            const u64 o = ((u64)L.khi[j] << 32) | L.klo[j];
            rank += (o > mine) ? 1 : 0; // keys are unique: (weight, slab index)
        }
        L.w[rank] = rw; L.mx[rank] = rmx; L.my[rank] = rmy;
        L.xx[rank] = rxx; L.xy[rank] = rxy; L.yy[rank] = ryy;
        const bool spd = (rxx > 0.f) && (ryy > 0.f) && (rxx * ryy - rxy * rxy > 0.f);
        L.tr[rank] = spd ? (rxx + ryy) : INFINITY;
        L.u[rank] = -1;
    }
    __syncthreads();
}

// sort 1 for the large mixtures without a sorting network: the weight keys are nearly uniform in their (orderable) bit
// pattern — a log scale — so a counting sort on the leading bits leaves buckets of a few survivors each, and a
// survivor's rank is its bucket's start plus the number of larger keys in its own bucket.  A quarter of the
// instructions of the register bitonic sort.  Returns false (uniformly, before anything the network needs is
// touched) when a bucket is crowded — many equal weights — and the network is the better tool.
__device__ __forceinline__ bool bucket_sort_survivors(const Lds& L, int S, int S_cap, int tid, int lane, int wave, int n_update)
{
    lds_u32 cntc = L.pay;             // per bucket: count -> (placed << 16) | start
    lds_u32 members = (lds_u32)L.u;   // bucket segments in arrival order (the slab indices live in the keys by then)
    const int NB = S_cap;             // buckets: a power of two >= 512
    u32 mh[4], ml[4];
    int bk[4];
    u32 kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        mh[e] = 0u; ml[e] = 0u; bk[e] = 0;
        if (i < S) {
            const int u0 = L.u[i];
            mh[e] = orderable(L.w[i]);
            ml[e] = 0xFFFFFFFFu - (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0);
            L.khi[i] = mh[e]; L.klo[i] = ml[e];
            kmn = mh[e] < kmn ? mh[e] : kmn;
            kmx = mh[e] > kmx ? mh[e] : kmx;
        }
    }
    for (int b = tid; b < NB; b += PHD_T) cntc[b] = 0u;
    // workgroup range of the weight keys
    {
        u32 o;
        o = xor_lane_c<32>(kmn); kmn = o < kmn ? o : kmn; o = xor_lane_c<32>(kmx); kmx = o > kmx ? o : kmx;
        o = xor_lane_c<16>(kmn); kmn = o < kmn ? o : kmn; o = xor_lane_c<16>(kmx); kmx = o > kmx ? o : kmx;
        o = xor_lane_c<8>(kmn); kmn = o < kmn ? o : kmn; o = xor_lane_c<8>(kmx); kmx = o > kmx ? o : kmx;
        o = xor_lane_c<4>(kmn); kmn = o < kmn ? o : kmn; o = xor_lane_c<4>(kmx); kmx = o > kmx ? o : kmx;
        o = xor_lane_c<2>(kmn); kmn = o < kmn ? o : kmn; o = xor_lane_c<2>(kmx); kmx = o > kmx ? o : kmx;
        o = xor_lane_c<1>(kmn); kmn = o < kmn ? o : kmn; o = xor_lane_c<1>(kmx); kmx = o > kmx ? o : kmx;
    }
    if (lane == 0) { L.ctr[CTR_TMP + wave] = (int)kmn; L.ctr[CTR_TMP + PHD_NW + wave] = (int)kmx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < PHD_NW; ++w) {
        const u32 a = (u32)L.ctr[CTR_TMP + w], b = (u32)L.ctr[CTR_TMP + PHD_NW + w];
        kmn = a < kmn ? a : kmn;
        kmx = b > kmx ? b : kmx;
    }
    const u32 range = kmx - kmn;
    const int bits = range ? 32 - __clz((int)range) : 0, lognb = 31 - __clz(NB);
    const int shift = bits > lognb ? bits - lognb : 0;          // (range >> shift) < NB; bucket 0 holds the largest weights
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        if (i < S) { bk[e] = (int)((kmx - mh[e]) >> shift); atomicAdd((u32*)&cntc[bk[e]], 1u); }
    }
    __syncthreads();
    // exclusive scan over the buckets (thread t owns buckets [t per, (t + 1) per)), and the largest bucket
    {
        const int per = NB / PHD_T;                               // 1, 2 or 4
        const int lo = tid * per;
        u32 v[4], local = 0u, lmax = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (e < per) ? cntc[lo + e] : 0u;
            local += v[e];
            lmax = v[e] > lmax ? v[e] : lmax;
        }
        const u32 incl = wave_incl_scan(local);
        lmax = (u32)wave_max_i((int)lmax);
        if (lane == 63) L.ctr[CTR_TMP + wave] = (int)incl;
        if (lane == 0) L.ctr[CTR_TMP + PHD_NW + wave] = (int)lmax;
        __syncthreads();
        u32 woff = 0u, bmax = 0u;
#pragma unroll
        for (int w = 0; w < PHD_NW; ++w) {
            const u32 c = (u32)L.ctr[CTR_TMP + w], m = (u32)L.ctr[CTR_TMP + PHD_NW + w];
            if (w < wave) woff += c;
            bmax = m > bmax ? m : bmax;
        }
        if (bmax > 128u) { __syncthreads(); return false; }       // uniform
        u32 run = woff + incl - local;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < per) { cntc[lo + e] = run; run += v[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        if (i < S) {
            const u32 old = atomicAdd((u32*)&cntc[bk[e]], 0x10000u);
            members[(old & 0xFFFFu) + (old >> 16)] = (u32)i;
        }
    }
    __syncthreads();
    int rank[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        rank[e] = 0;
        if (i < S) {
            const u32 c = cntc[bk[e]];
            const int base = (int)(c & 0xFFFFu), k = (int)(c >> 16);
            const u64 mine = ((u64)mh[e] << 32) | ml[e];
            int r = 0;
            for (int t = 0; t < k; ++t) {
                const int m = (int)members[base + t];
                const u64 o = ((u64)L.khi[m] << 32) | L.klo[m];
                r += (o > mine) ? 1 : 0;                           // keys are unique: (weight, slab index)
            }
            rank[e] = base + r;
        }
    }
    // the planes are staged through registers only now (short live ranges: the kernel sits at its register budget)
    float rw[4], rmx[4], rmy[4], rxx[4], rxy[4], ryy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        rw[e] = rmx[e] = rmy[e] = rxx[e] = rxy[e] = ryy[e] = 0.f;
        if (i < S) { rw[e] = L.w[i]; rmx[e] = L.mx[i]; rmy[e] = L.my[i]; rxx[e] = L.xx[i]; rxy[e] = L.xy[i]; ryy[e] = L.yy[i]; }
    }
    __syncthreads(); // every read of the old order (and of the member lists, which sit in u) precedes the writes below
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        if (i < S) {
            const int k = rank[e];
            L.w[k] = rw[e]; L.mx[k] = rmx[e]; L.my[k] = rmy[e];
            L.xx[k] = rxx[e]; L.xy[k] = rxy[e]; L.yy[k] = ryy[e];
            const bool spd = (rxx[e] > 0.f) && (ryy[e] > 0.f) && (rxx[e] * ryy[e] - rxy[e] * rxy[e] > 0.f);
            L.tr[k] = spd ? (rxx[e] + ryy[e]) : INFINITY;
            L.u[k] = -1; // unassigned
        }
    }
    __syncthreads();
    return true;
}

// Is survivor e within the merge distance of seed s?  (d(s,e) < T, src/phdfilter.cu:2802-2806)
//
// Mahalanobis: the decision is taken WITHOUT the four divisions of the reference formula whenever it
// is not marginal.  With s0,s1,s3,det,d0,d1 computed exactly as mahal_dist() computes them (FMA
// contraction off -> bitwise the same values), the reference's d differs from q/det,
// q = d0^2 s3 - 2 d0 d1 s1 + d1^2 s0, by at most ~6 ulp of A/det, A = sum of |terms|; q itself is
// computed here with the same bound.  So |q - T det| > 4e-6 (A + T det)  (67 ulp) decides the
// comparison d < T exactly as the reference formula would; inside that band — or if det <= 0 or
// anything is non-finite — the reference formula itself is evaluated.
template <bool HELLINGER>
__device__ __forceinline__ bool is_close(float smx, float smy, float sxx, float sxy, float syy,
                                         float emx, float emy, float exx, float exy, float eyy, float T)
{
#pragma clang fp contract(off)
    if (HELLINGER) return hellinger_dist(smx, smy, sxx, sxy, syy, emx, emy, exx, exy, eyy) < T;
    const float s0 = (sxx + exx) * 0.5f;
    const float s1 = (sxy + exy) * 0.5f;
    const float s3 = (syy + eyy) * 0.5f;
    const float det = s0 * s3 - s1 * s1;
    const float d0 = smx - emx;
    const float d1 = smy - emy;
    const float t1 = d0 * d0 * s3, t2 = d0 * d1 * s1, t3 = d1 * d1 * s0;
    const float q = t1 - 2.f * t2 + t3;
    const float A = fabsf(t1) + 2.f * fabsf(t2) + fabsf(t3);
    const float Td = T * det;
    const float tol = 4e-6f * (A + fabsf(Td));
    if (det > 0.f && T > 0.f) {
        const float diff = q - Td;
        if (diff > tol) return false;
        if (diff < -tol) return true;
    }
    return mahal_dist(smx, smy, sxx, sxy, syy, emx, emy, exx, exy, eyy) < T;
}

// ------------------------------------------------------------------------------------------
// merge_small: the same greedy merge for S <= 256 survivors in ONE shot instead of rounds.
//
// With at most 256 survivors every position is a candidate seed, so the whole decision structure fits in
// 256-bit masks:  rank (counting, the idle threads share the key scan) -> planes permuted into
// (weight desc, slab index asc) order -> row_k = {l < k : close(k, l)} for all pairs (wave = 64 positions x a
// 128-column chunk, column data by LDS broadcast; cheap trace filter, then the exact decision) ->
// seeds s_k = not exists l < k : close(k,l) and s_l, resolved by one wave, 64 positions at a time (earlier
// blocks are final, inside a block the ballot fixed point of the round-based version) -> every position
// joins the first seed of its row (LDS atomic OR into the seed's member mask) -> the thread that owns a seed
// walks its members in ascending position = (weight desc) order and does the moment matching
// (src/gm_reduce.cpp:103-118's order), including the reference's stop rule (src/phdfilter.cu:2821).
// No second sort, no segment pass, no list compaction: 6 barriers instead of ~25.
// ------------------------------------------------------------------------------------------
template <bool HELLINGER, bool STAMPS>
__device__ __forceinline__ void merge_small(const Lds& L, int S_cap, int S, const DevConfig& cfg, float* __restrict__ out_slab,
                                            int cap, int tid, u64* st, int n_update)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const float T = cfg.minSeparation;
    const float Tpre = (T > 0.f) ? T * 1.01f : -1.f;
    // ---- rank by counting; nh = PHD_T / S (at most 8) threads per survivor share the scan of the keys: thread
    //      tid serves survivor tid mod S, so that nearly every lane of the workgroup has work whatever S is
    const int nh = (PHD_T / S) < 8 ? (PHD_T / S) : 8;   // S <= 256: at least 2
    const int h = tid / S, i = tid - h * S;
    // the survivor planes are dead once the sorted staging has read them: the last three (yy, tr, u: >= 3 KB) hold the
    // filter's packed copy of the sorted means and radii
    lds_f32 fX2 = L.yy;                        // [128] (mx_l, mx_l+1, my_l, my_l+1)
    lds_f32 fZ2 = L.yy + 2 * PHD_SMALL_S;      // [256] 0.505 T tr_l
    LDS_T(u64)* skey = L.srow;                 // [256] 64-bit keys (the row area is free until the ranks are known)
    lds_u32 scnt = (lds_u32)(L.srow + PHD_SMALL_S);
    u32 mh = 0, ml = 0;
    float rw = 0, rmx = 0, rmy = 0, rxx = 0, rxy = 0, ryy = 0;
    if (h == 0) {
        const int u0 = L.u[i];
        mh = orderable(L.w[i]);
        ml = 0xFFFFFFFFu - (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0);
        skey[i] = ((u64)mh << 32) | ml;
        scnt[i] = 0u;
        rw = L.w[i]; rmx = L.mx[i]; rmy = L.my[i]; rxx = L.xx[i]; rxy = L.xy[i]; ryy = L.yy[i];
    }
    __syncthreads();
    if (h < nh) {
        const u64 mine = skey[i];
        const int per = (S + nh - 1) / nh;
        const int j0 = h * per, j1 = (j0 + per < S) ? j0 + per : S;
        int cnt = 0;
#pragma unroll 8
        for (int j = j0; j < j1; ++j) cnt += (skey[j] > mine) ? 1 : 0; // keys are unique: (weight, slab index)
        atomicAdd((u32*)&scnt[i], (u32)cnt);
    }
    __syncthreads();
    if (h == 0) {
        const int rank = (int)scnt[i];
        const bool spd = (rxx > 0.f) && (ryy > 0.f) && (rxx * ryy - rxy * rxy > 0.f);
        const float tr = spd ? (rxx + ryy) : INFINITY;
        // the sorted order lives in two float4 arrays (the SoA planes keep the arrival order): one ds_read_b128 per
        // column in the filter, two per operand in the exact test and the moment matching
        L.sA[rank] = (v4f){rmx, rmy, 0.5f * Tpre * tr, rw};
        L.sB[rank] = (v4f){rxx, rxy, ryy, 0.f};
        // the filter's copy, two columns per entry so that its arithmetic is packed (v_pk_*): (mx, mx', my, my'), (z, z')
        fX2[(rank >> 1) * 4 + (rank & 1)] = rmx;
        fX2[(rank >> 1) * 4 + 2 + (rank & 1)] = rmy;
        fZ2[rank] = 0.5f * Tpre * tr;
    }
    __syncthreads();
    STAMP(6);
    u64 tq0 = 0, tq1 = 0, tq2 = 0;
    lds_u32 plist = (lds_u32)L.w;                // candidate pairs (k << 16 | l): the first five survivor planes
    const int pcap = 5 * S_cap;
    // ---- closeness rows.  Work units are HALF waves: 32 consecutive positions (half-block hb) x 16 columns (column unit
    //      cu, columns 16 cu .. 16 cu + 15, only units with a column below the half-block's last row), two units per
    //      wave iteration, dealt round-robin to the waves; a unit fills the 16-bit quarter cu of its rows' words.
    //      (With 64-position units a 148-survivor mixture paid for 24 units of which 44 % was real work; now 15.)
    const int ncu = (S + 15) >> 4, nhb = (S + 31) >> 5;
    int n_half = 0;
    for (int hb = 0; hb < nhb; ++hb) n_half += (2 * hb + 2 < ncu) ? 2 * hb + 2 : ncu;
    {
        static_assert(PHD_SMALL_S == 256, "merge_small uses four 64-bit words per row");
        // (the rows are not cleared: every quarter a later phase reads — words lc <= k / 64 of the rows k < S — is written
        //  below; the member masks are cleared here, they are first touched two barriers later)
        for (int t = tid; t < PHD_SMALL_S * 4; t += PHD_T) L.scol[t] = 0ull;
        if (STAMPS && tid == 0) tq0 = __builtin_amdgcn_s_memrealtime();
        LDS_T(u16)* srow16 = (LDS_T(u16)*)L.srow;
        for (int u0 = 2 * wave; u0 < n_half; u0 += 2 * PHD_NW) {
            const int h = u0 + (lane >> 5);
            const bool hvalid = h < n_half;
            int hb = 0, cu = hvalid ? h : 0;
            for (;;) { const int c = (2 * hb + 2 < ncu) ? 2 * hb + 2 : ncu; if (cu < c) break; cu -= c; ++hb; }
            const int cnt_hb = (2 * hb + 2 < ncu) ? 2 * hb + 2 : ncu;
            const int lbase = 16 * cu;
            const int k = 32 * hb + (lane & 31);
            const bool kvalid = hvalid && k < S;
            const int kk = (k < S) ? k : S - 1;
            const v4f ka = L.sA[kk];
            const float kmx = ka.x, kmy = ka.y, kat = ka.z;
            // cheap conservative filter over the unit's 16 columns, branch-free: d^2 < 0.505 T (tr_l + tr_k)
            // (d >= 2|dm|^2/(tr Pa + tr Pb) for SPD covariances, 1 % guard band; +inf trace = "always a candidate").
            // The column data are LDS broadcast reads (one address per half wave), all in flight.
            // Columns >= S hold stale data of earlier steps: the test l < k (< S) masks them.
            u32 cand = 0;
            const LDS_T(v4f)* cX = (const LDS_T(v4f)*)fX2 + (lbase >> 1);
            const LDS_T(v2f)* cZ = (const LDS_T(v2f)*)fZ2 + (lbase >> 1);
            const v2f kx2 = (v2f){kmx, kmx}, ky2 = (v2f){kmy, kmy}, kz2 = (v2f){kat, kat};
#pragma unroll
            for (int jp = 0; jp < 8; ++jp) {
                const v4f a = cX[jp];
                const v2f z = cZ[jp];
                const v2f dx = (v2f){a.x, a.y} - kx2, dy = (v2f){a.z, a.w} - ky2;
                const v2f d2 = dx * dx + dy * dy, thr = z + kz2;
                const bool near0 = HELLINGER || !(d2.x >= thr.x), near1 = HELLINGER || !(d2.y >= thr.y);
                cand |= (near0 ? (1u << (2 * jp)) : 0u) | (near1 ? (2u << (2 * jp)) : 0u);
            }
            // only earlier positions count (l < k), and only rows of real positions
            const int nlt = k - lbase;
            cand &= (!kvalid || nlt <= 0) ? 0u : (nlt >= 16 ? 0xFFFFu : ((1u << nlt) - 1u));
            // The marked pairs are few (a hundred or two per particle) and unevenly spread over the positions, so the
            // exact decision does not run here, one divergent loop per lane: the row keeps the candidate bits and the
            // pairs go to a list (wave-aggregated slot allocation) that the whole workgroup tests one pair per
            // thread below.  The list lives in the survivor planes, dead since the sorted staging.
            if (hvalid) {
                srow16[k * 16 + cu] = (u16)cand;
                // the unit on the first quarter of the rows' own word also zeroes the quarters of that word no unit covers
                const int q0 = 4 * (hb >> 1);
                if (cu == q0) {
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                        if (q0 + q >= cnt_hb) srow16[k * 16 + q0 + q] = 0;
                }
            }
            const int np = __popc(cand);
            const int incl = (int)wave_incl_scan((u32)np);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot) {
                int base = 0;
                if (lane == 63) base = atomicAdd((int*)&L.ctr[CTR_NPAIR], tot);
                int pos = __builtin_amdgcn_readlane(base, 63) + incl - np;
                while (cand) {
                    const int j = __builtin_ctz(cand);
                    cand &= cand - 1;
                    if (pos < pcap) plist[pos] = ((u32)k << 16) | (u32)(lbase + j);
                    ++pos;
                }
            }
        }
    }
    __syncthreads();
    {
        const int n_pairs = L.ctr[CTR_NPAIR];
        if (n_pairs <= pcap) {
            // exact decision, one listed pair per thread: a candidate that fails loses its bit
            for (int t = tid; t < n_pairs; t += PHD_T) {
                const u32 pr = plist[t];
                const int k = (int)(pr >> 16), l = (int)(pr & 0xFFFFu);
                const v4f ka = L.sA[k], kbv = L.sB[k], la = L.sA[l], lb = L.sB[l];
                if (!is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, ka.x, ka.y, kbv.x, kbv.y, kbv.z, T))
                    atomicAnd((u64*)&L.srow[k * 4 + (l >> 6)], ~(1ull << (l & 63)));
            }
        } else {
            // more candidates than the list holds (dense clutter of overlapping Gaussians, or the Hellinger metric, which
            // has no cheap filter): the exact decision per position on its marked columns
            LDS_T(u16)* srow16 = (LDS_T(u16)*)L.srow;
            for (int u0 = 2 * wave; u0 < n_half; u0 += 2 * PHD_NW) {
                const int h = u0 + (lane >> 5);
                if (h >= n_half) continue;
                int hb = 0, cu = h;
                for (;;) { const int c = (2 * hb + 2 < ncu) ? 2 * hb + 2 : ncu; if (cu < c) break; cu -= c; ++hb; }
                const int lbase = 16 * cu;
                const int k = 32 * hb + (lane & 31);
                const int kk = k < S ? k : S - 1;
                const v4f ka = L.sA[kk], kbv = L.sB[kk];
                u32 cand = srow16[k * 16 + cu];
                u32 bits = 0;
                while (cand) {
                    const int j = __builtin_ctz(cand);
                    cand &= cand - 1;
                    const int l = lbase + j;
                    const v4f la = L.sA[l], lb = L.sB[l];
                    if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, ka.x, ka.y, kbv.x, kbv.y, kbv.z, T)) bits |= 1u << j;
                }
                srow16[k * 16 + cu] = (u16)bits;
            }
        }
    }
    __syncthreads();
    if (STAMPS && tid == 0) tq1 = __builtin_amdgcn_s_memrealtime();
    // ---- seeds: one wave, block by block
    if (wave == 0) {
        u64 sd[4] = {0ull, 0ull, 0ull, 0ull};
        const int nblk = (S + 63) >> 6;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b < nblk) {
                const int k = 64 * b + lane;
                const bool kvalid = k < S;
                u64 r0 = L.srow[k * 4 + 0], r1 = L.srow[k * 4 + 1], r2 = L.srow[k * 4 + 2], r3 = L.srow[k * 4 + 3];
                // blocked by a (final) seed of an earlier block?
                bool blocked = false;
                if (b > 0) blocked = blocked || ((r0 & sd[0]) != 0ull);
                if (b > 1) blocked = blocked || ((r1 & sd[1]) != 0ull);
                if (b > 2) blocked = blocked || ((r2 & sd[2]) != 0ull);
                const u64 row = (b == 0) ? r0 : (b == 1) ? r1 : (b == 2) ? r2 : r3;   // within the block
                const u64 live = __ballot(kvalid && !blocked);
                u64 seeds = live;
                for (int it = 0; it < 65; ++it) {
                    const u64 blk = __ballot((row & seeds) != 0ull);
                    const u64 nx = live & ~blk;
                    if (nx == seeds) break;
                    seeds = nx;
                }
                sd[b] = seeds;
            }
        }
        if (lane < 4) L.sseed[lane] = (lane == 0) ? sd[0] : (lane == 1) ? sd[1] : (lane == 2) ? sd[2] : sd[3];
    }
    __syncthreads();
    if (STAMPS && tid == 0) tq2 = __builtin_amdgcn_s_memrealtime();
    // ---- membership: every position joins the first seed of its row (a seed joins itself)
    const u64 s0 = L.sseed[0], s1 = L.sseed[1], s2 = L.sseed[2], s3 = L.sseed[3];
    bool is_seed = false;
    if (tid < S) {
        const int k = tid;
        const u64 sw = (k < 64) ? s0 : (k < 128) ? s1 : (k < 192) ? s2 : s3;
        is_seed = (sw >> (k & 63)) & 1ull;
        int owner = k;
        if (!is_seed) {
            const int kbk = k >> 6;                  // words beyond the position's own block were never written
            const u64 m0 = L.srow[k * 4 + 0] & s0, m1 = (kbk >= 1) ? (L.srow[k * 4 + 1] & s1) : 0ull,
                      m2 = (kbk >= 2) ? (L.srow[k * 4 + 2] & s2) : 0ull, m3 = (kbk >= 3) ? (L.srow[k * 4 + 3] & s3) : 0ull;
            owner = m0 ? __builtin_ctzll(m0) : m1 ? 64 + __builtin_ctzll(m1) : m2 ? 128 + __builtin_ctzll(m2)
                                                                            : 192 + __builtin_ctzll(m3);
        }
        atomicOr((u64*)&L.scol[owner * 4 + (k >> 6)], 1ull << (k & 63));
    }
    if (tid == 0) L.ctr[CTR_KOUT] = 0x7FFFFFFF;
    __syncthreads();
    if (STAMPS && tid == 0) {
        const u64 tq3 = __builtin_amdgcn_s_memrealtime();
        st[12] += tq1 - tq0; st[13] += tq2 - tq1; st[14] += tq3 - tq2; st[15] += 1;
    }
    STAMP(7);
    STAMP(8);
    STAMP(9);
    // ---- moment matching: the thread that owns a seed, members in ascending position
    const int n_clusters = __popcll(s0) + __popcll(s1) + __popcll(s2) + __popcll(s3);
    if (is_seed) {
#pragma clang fp contract(off)
        const int k = tid;
        const u64 below = (k & 63) ? (~0ull >> (64 - (k & 63))) : 0ull;
        int c = 0; // cluster index = seeds before this one
        c += (k >= 64) ? __popcll(s0) : __popcll(s0 & below);
        if (k >= 64) c += (k >= 128) ? __popcll(s1) : __popcll(s1 & below);
        if (k >= 128) c += (k >= 192) ? __popcll(s2) : __popcll(s2 & below);
        if (k >= 192) c += __popcll(s3 & below);
        u64 mem[4] = {L.scol[k * 4 + 0], L.scol[k * 4 + 1], L.scol[k * 4 + 2], L.scol[k * 4 + 3]};
        const v4f sa = L.sA[k], sb = L.sB[k];
        const float smx = sa.x, smy = sa.y, sxx = sb.x, sxy = sb.y, syy = sb.z;
        const float dself = HELLINGER ? hellinger_dist(smx, smy, sxx, sxy, syy, smx, smy, sxx, sxy, syy)
                                      : mahal_dist(smx, smy, sxx, sxy, syy, smx, smy, sxx, sxy, syy);
        const bool selfok = dself < T;
        if (!selfok) mem[k >> 6] &= ~(1ull << (k & 63)); // a seed that is not close to itself is not in its own cluster
        float W = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            u64 m = mem[wd];
            while (m) {
                const int p = 64 * wd + __builtin_ctzll(m);
                m &= m - 1;
                const v4f pa = L.sA[p];
                const float w = pa.w;
                W += w;
                sx += w * pa.x;
                sy += w * pa.y;
            }
        }
        // reference loop: W == 0 -> break (src/phdfilter.cu:2821); a seed left unmerged is re-picked and then
        // yields W == 0
        int stop_at = 0x7FFFFFFF;
        if (W == 0.f) stop_at = c;
        else if (!selfok) stop_at = c + 1;
        if (stop_at != 0x7FFFFFFF) atomicMin((int*)&L.ctr[CTR_KOUT], stop_at);
        if (W != 0.f && c < cap) {
            const float mx = sx / W, my = sy / W;
            float cxx = 0.f, cxy = 0.f, cyy = 0.f;
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                u64 m = mem[wd];
                while (m) {
                    const int p = 64 * wd + __builtin_ctzll(m);
                    m &= m - 1;
                    const v4f pa = L.sA[p], pb = L.sB[p];
                    const float w = pa.w;
                    const float d0 = mx - pa.x;
                    const float d1 = my - pa.y;
                    cxx += w * (pb.x + d0 * d0);
                    cxy += w * (pb.y + d0 * d1);
                    cyy += w * (pb.z + d1 * d1);
                }
            }
            out_slab[0 * cap + c] = W;
            out_slab[1 * cap + c] = mx;
            out_slab[2 * cap + c] = my;
            out_slab[3 * cap + c] = cxx / W;
            out_slab[4 * cap + c] = cxy / W;
            out_slab[5 * cap + c] = cyy / W;
        }
    }
    __syncthreads();
    STAMP(10);
    if (tid == 0) {
        int k = L.ctr[CTR_KOUT];
        if (k > n_clusters) k = n_clusters;
        L.ctr[CTR_KOUT] = k;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// the greedy merge on the survivors held in LDS; writes the merged map to the output slab.
// Returns (in ctr[CTR_KOUT]) the number of merged Gaussians.
// ------------------------------------------------------------------------------------------
template <bool HELLINGER, bool STAMPS>
__device__ __forceinline__ void merge_in_lds(const Lds& L, int S_cap, int n_surv, const DevConfig& cfg, float* __restrict__ out_slab,
                             int cap, int tid, u64* st, int n_update, bool packed)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const float T = cfg.minSeparation;
    // guard band of the trace filter; T <= 0: "2 d2 >= -(..)" always holds -> far, like the exact test
    const float Tpre = (T > 0.f) ? T * 1.01f : -1.f;
    const int S = n_surv;
    if (tid == 0) { L.ctr[CTR_KOUT] = 0; L.ctr[CTR_NHEAD] = 0; }
    if (S == 0) { __syncthreads(); return; }
    if (S <= PHD_SMALL_S) { merge_small<HELLINGER, STAMPS>(L, S_cap, S, cfg, out_slab, cap, tid, st, n_update); return; }

    // ---- sort 1: (weight desc, slab index asc) ------------------------------------------------
    int n_pad = 2;
    while (n_pad < S) n_pad <<= 1;
    if (bucket_sort_survivors(L, S, S_cap, tid, lane, wave, n_update)) { /* counting sort did it */ }
    else if (n_pad <= PHD_T) rank_sort_survivors(L, S, tid, n_update);
    else if (n_pad <= 2 * PHD_T) sort_survivors<2>(L, S, n_pad, tid, n_update, packed);
    else if (n_pad <= 4 * PHD_T) sort_survivors<4>(L, S, n_pad, tid, n_update, packed);
    else sort_survivors<8>(L, S, n_pad, tid, n_update, packed);
    STAMP(6);

    // ---- rounds: 64 live candidates at a time ---------------------------------------------------
    // `cur` lists the still-unmerged survivors in (weight desc) order.  Per round the first 64 of
    // them form the window: their pairwise closeness decides which are seeds (a candidate is a seed
    // iff no earlier seed is close to it), every later survivor is assigned to the first seed it is
    // close to, and the list is compacted.  Exactly the reference's greedy loop, 64 seeds at a time.
    lds_i32 assign = L.u;
    lds_u16 ul_a = (lds_u16)L.pay, ul_b = ul_a + S_cap;  // two u16 lists in the (free) sort-payload region
    LDS_T(u64)* cmask = (LDS_T(u64)*)L.khi;              // candidate-seed mask per listed survivor (khi+klo)
#if PHD_NW == 4
    typedef u16 cmw_t;
#else
    typedef unsigned char cmw_t;
#endif
    LDS_T(cmw_t)* cmw = (LDS_T(cmw_t)*)L.khi;
    lds_i32 wpos = (lds_i32)L.win;
    lds_f32 wmx = L.win + 64, wmy = L.win + 128, wtr = L.win + 192, wxx = L.win + 256, wxy = L.win + 320,
            wyy = L.win + 384;
    for (int i = tid; i < S; i += PHD_T) ul_a[i] = (u16)i;
    int n_u = S;
    lds_u16 cur = ul_a, nxt = ul_b;
    __syncthreads();
    while (n_u > 0) {
        u64 tq0 = 0, tq1 = 0, tq2 = 0;
        if (STAMPS && tid == 0) tq0 = __builtin_amdgcn_s_memrealtime();
        const int nwin = n_u < 64 ? n_u : 64;
        const int nrest = n_u - nwin;
        // (0) window buffer: broadcast-friendly copy of the candidates
        if (tid < 64) {
            const int i = cur[tid < nwin ? tid : nwin - 1];
            wpos[tid] = i;
            wmx[tid] = L.mx[i]; wmy[tid] = L.my[i]; wtr[tid] = (tid < nwin) ? L.tr[i] : -INFINITY;
            wxx[tid] = L.xx[i]; wxy[tid] = L.xy[i]; wyy[tid] = L.yy[i];
        }
        __syncthreads();
        // (1) closeness matrix rows: lane = candidate k, wave = column block [COLS*wave, COLS*(wave+1)).
        //     A cheap conservative filter (d >= 2|dm|^2/(tr Pa + tr Pb) for SPD covariances, 1 % guard
        //     band) marks candidate columns; the exact test runs on the marked bits only.
        {
            const int k = lane;
            const bool kvalid = k < nwin;
            const float kmx = wmx[k], kmy = wmy[k], ktr = wtr[k];
            u32 cand = 0;
#pragma unroll 4
            for (int c = 0; c < PHD_COLS; ++c) {
                const int l = wave * PHD_COLS + c;
                const float dx = wmx[l] - kmx, dy = wmy[l] - kmy;
                const bool near = HELLINGER || !(2.f * (dx * dx + dy * dy) >= Tpre * (wtr[l] + ktr));
                if (kvalid && l < k && near) cand |= (1u << c);
            }
            u32 bits = 0;
            if (cand) {
                const float kxx = wxx[k], kxy = wxy[k], kyy = wyy[k];
                while (cand) {
                    const int c = __builtin_ctz(cand);
                    cand &= cand - 1;
                    const int l = wave * PHD_COLS + c;
                    if (is_close<HELLINGER>(wmx[l], wmy[l], wxx[l], wxy[l], wyy[l], kmx, kmy, kxx, kxy, kyy, T))
                        bits |= (1u << c);
                }
            }
            L.part[wave * 64 + k] = bits;
        }
        __syncthreads();
        if (STAMPS && tid == 0) tq1 = __builtin_amdgcn_s_memrealtime();
        // (2) seeds: s_k = not exists l < k : close(k,l) and s_l.  The recursion is well founded, so the
        //     parallel iteration s <- F(s) reaches its unique fixed point (position k is final after k+1
        //     sweeps; in practice a handful).  Every wave computes the same mask.
        u64 seeds;
        {
            u64 row = 0;
#pragma unroll
            for (int wv = 0; wv < PHD_NW; ++wv) row |= (u64)L.part[wv * 64 + lane] << (PHD_COLS * wv);
            const u64 live = (nwin == 64) ? ~0ull : ((1ull << nwin) - 1ull);
            seeds = live;
            for (int it = 0; it < 65; ++it) {
                const u64 blocked = __ballot((row & seeds) != 0ull);
                const u64 nx = live & ~blocked;
                if (nx == seeds) break;
                seeds = nx;
            }
            if (wave == 0 && lane < nwin) {
                const int owner = ((seeds >> lane) & 1ull) ? lane : __builtin_ctzll(row & seeds);
                assign[wpos[lane]] = wpos[owner];
            }
        }
        if (STAMPS && tid == 0) tq2 = __builtin_amdgcn_s_memrealtime();
        // (3a) cheap filter for the survivors after the window: wave w owns the window's candidates
        //      [COLS*w, COLS*(w+1)) and sweeps all listed survivors, 64 (one per lane) at a time ->
        //      COLS candidate-seed bits per (survivor, wave), stored as one field of the survivor's u64
        {
            const u32 myseeds = (u32)(seeds >> (PHD_COLS * wave)) & ((1u << PHD_COLS) - 1u);
            // the wave's candidates are the same for the whole sweep: read them once and keep them in scalar registers
            // (the sweep is bound by LDS return bandwidth — a broadcast read still returns a full wave of data)
            float smx[PHD_COLS], smy[PHD_COLS], str[PHD_COLS];
#pragma unroll
            for (int c = 0; c < PHD_COLS; ++c) {
                const int l = PHD_COLS * wave + c;
                smx[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wmx[l])));
                smy[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wmy[l])));
                // the filter's right-hand side Tpre (tr_l + tr_e) / 2 is split into a per-candidate and a per-survivor half
                // (it is a conservative bound with a 1 % guard band: the rounding of the split is immaterial)
                str[c] = 0.5f * Tpre * __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wtr[l])));
            }
            for (int e0 = 0; e0 < nrest; e0 += 64) {
                const int e = e0 + lane;
                u32 mbits = 0;
                if (myseeds) {
                    float emx = 0, emy = 0, etr = -INFINITY; // -inf trace: the filter rejects
                    if (e < nrest) { const int i = cur[64 + e]; emx = L.mx[i]; emy = L.my[i]; etr = L.tr[i]; }
                    const float eth = 0.5f * Tpre * etr;
#pragma unroll
                    for (int g = 0; g < PHD_COLS / 4; ++g) {
                        if (!((myseeds >> (4 * g)) & 0xFu)) continue; // uniform
#pragma unroll
                        for (int q = 0; q < 4; q += 2) { // two candidates per step: packed arithmetic
                            const int c = 4 * g + q;
                            const v2f dx = (v2f){smx[c], smx[c + 1]} - (v2f){emx, emx};
                            const v2f dy = (v2f){smy[c], smy[c + 1]} - (v2f){emy, emy};
                            const v2f lhs = dx * dx + dy * dy;
                            const v2f rhs = (v2f){str[c], str[c + 1]} + (v2f){eth, eth};
                            const bool near0 = HELLINGER ? (etr > -INFINITY) : !(lhs.x >= rhs.x);
                            const bool near1 = HELLINGER ? (etr > -INFINITY) : !(lhs.y >= rhs.y);
                            mbits |= (near0 ? (1u << c) : 0u) | (near1 ? (2u << c) : 0u);
                        }
                    }
                    mbits &= myseeds;
                }
                if (e < nrest) cmw[e * PHD_NW + wave] = (cmw_t)mbits;
            }
        }
        __syncthreads();
        // (3b) exact test on the candidates only, in seed order, until the first hit; then the ordered
        //      compaction of the list (thread t owns the contiguous entries [t*per, (t+1)*per))
        {
            const int per = (nrest + PHD_T - 1) / PHD_T;
            const int e_lo = tid * per, e_hi = (e_lo + per < nrest) ? e_lo + per : nrest;
            int kept = 0;
            for (int e = e_lo; e < e_hi; ++e) {
                const int i = cur[64 + e];
                u64 m = cmask[e];
                bool merged = false;
                if (m) {
                    const float fmx = L.mx[i], fmy = L.my[i], fxx = L.xx[i], fxy = L.xy[i], fyy = L.yy[i];
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        if (is_close<HELLINGER>(wmx[l], wmy[l], wxx[l], wxy[l], wyy[l], fmx, fmy, fxx, fxy, fyy, T)) {
                            assign[i] = wpos[l];
                            merged = true;
                            break;
                        }
                    }
                }
                if (!merged) kept++;
                cmask[e] = merged ? 0ull : 1ull; // reuse as the keep flag of the compaction
            }
            // exclusive scan of `kept` over the workgroup (wave shuffle scan + wave totals)
            const int incl = (int)wave_incl_scan((u32)kept);
            if (lane == 63) L.ctr[CTR_TMP + wave] = incl;
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < PHD_NW; ++w) {
                const int c = L.ctr[CTR_TMP + w];
                if (w < wave) woff += c;
                total += c;
            }
            int o = woff + incl - kept;
            for (int e = e_lo; e < e_hi; ++e)
                if (cmask[e]) nxt[o++] = cur[64 + e];
            n_u = total;
            __syncthreads();
            lds_u16 t2 = cur; cur = nxt; nxt = t2;
        }
        if (STAMPS && tid == 0) {
            const u64 tq3 = __builtin_amdgcn_s_memrealtime();
            st[12] += tq1 - tq0; st[13] += tq2 - tq1; st[14] += tq3 - tq2; st[15] += 1;
        }
    }

    STAMP(7);
    // ---- group by seed, members in sorted-position order --------------------------------------------
    // A counting sort instead of a second bitonic sort (a tenth of its instructions): members per seed (LDS atomics)
    // -> one packed scan gives each seed its segment start (low half) and its cluster index (high half: seeds
    // before it) -> members dropped into their seed's segment in arrival order -> every member finds its place by
    // counting the smaller positions in its own segment (clusters are small), which makes the order — and so the
    // summation order of the moment matching — deterministic.
    {
        lds_u32 cursor = L.khi;                 // per seed: members placed so far (then key2, the grouped list)
        lds_u32 members = L.klo;                // segments in arrival order (then seg, the cluster starts)
        lds_u32 cnt = L.pay;                    // per seed: member count -> packed exclusive prefix
        for (int i = tid; i < S; i += PHD_T) { cnt[i] = 0u; cursor[i] = 0u; }
        __syncthreads();
        for (int i = tid; i < S; i += PHD_T) atomicAdd((u32*)&cnt[assign[i]], 1u);
        __syncthreads();
        const int per = (S + PHD_T - 1) / PHD_T; // <= 4 (S <= 2048)
        u32 total;
        {
            const int lo = tid * per;
            u32 v[4], local = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sd = lo + e;
                v[e] = (e < per && sd < S) ? (cnt[sd] | ((assign[sd] == sd) ? 0x10000u : 0u)) : 0u;
                local += v[e];
            }
            const u32 incl = wave_incl_scan(local);
            if (lane == 63) L.ctr[CTR_TMP + wave] = (int)incl;
            __syncthreads();
            u32 woff = 0u;
            total = 0u;
#pragma unroll
            for (int w = 0; w < PHD_NW; ++w) {
                const u32 c = (u32)L.ctr[CTR_TMP + w];
                if (w < wave) woff += c;
                total += c;
            }
            u32 run = woff + incl - local;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sd = lo + e;
                if (e < per && sd < S) { cnt[sd] = run; run += v[e]; }
            }
        }
        __syncthreads();
        int sreg[4], pos[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + e * PHD_T;
            sreg[e] = 0;
            if (i < S) {
                const int sd = assign[i];
                sreg[e] = sd;
                members[(cnt[sd] & 0xFFFFu) + atomicAdd((u32*)&cursor[sd], 1u)] = (u32)i;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + e * PHD_T;
            pos[e] = 0;
            if (i < S) {
                const int b = (int)(cnt[sreg[e]] & 0xFFFFu), k = (int)cursor[sreg[e]];
                int r = 0;
                for (int t = 0; t < k; ++t) r += ((int)members[b + t] < i) ? 1 : 0;
                pos[e] = b + r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + e * PHD_T;
            if (i < S) {
                L.key2[pos[e]] = ((u32)sreg[e] << 16) | (u32)i;                    // key2 aliases cursor
                if (sreg[e] == i) L.seg[cnt[i] >> 16] = (int)(cnt[i] & 0xFFFFu);   // seg aliases members
            }
        }
        if (tid == 0) { L.seg[total >> 16] = S; L.ctr[CTR_NHEAD] = (int)(total >> 16); L.ctr[CTR_KOUT] = 0x7FFFFFFF; }
        __syncthreads();
    }
    STAMP(8);
    const int n_clusters = L.ctr[CTR_NHEAD];
    STAMP(9);

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
    STAMP(10);
    if (tid == 0) {
        int k = L.ctr[CTR_KOUT];
        if (k > n_clusters) k = n_clusters;
        L.ctr[CTR_KOUT] = k;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// vehicle predict (phdPredictKernelAckerman, src/phdfilter.cu:785-825) — shared by the stand-alone
// predict kernel and the fused step
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// counter-based generator: Box-Muller on two splitmix64 outputs (replaces rng.cpp's wall-clock
// seeded boost::mt19937; draw order (n_alpha, n_encoder), src/phdfilter.cu:1148-1152)
__device__ __forceinline__ void draw_noise(u64 seed, u64 counter, int i, const DevConfig& cfg, float& n_alpha,
                                           float& n_encoder)
{
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

__device__ __forceinline__ phd_pose predict_pose(const phd_pose& o, phd_ackerman_control u, float n_alpha,
                                                 float n_encoder, const DevConfig& cfg)
{
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
    return nw;
}

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

size_t cphd_lds_bytes(int cn_len, int MM)
{
    u32 off[10];
    return cphd_lds_layout(cn_len, MM, off);
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

// the two sweeps of cphd_block for a compile-time number of 64-lane tiles (M <= 64 TILES): with TILES = 1 — every
// configuration of BASELINE.json — the tile loops and their guards fold away, which halves the instruction count
// of this single-wave, latency-bound section
template <int tiles>
__device__ __forceinline__ void cphd_esf_backward(const CphdLds& Q, float2* __restrict__ T_scratch, int M, int lane, float llam,
                                                  float lam)
{
#pragma clang fp contract(off)
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
            T_scratch[(size_t)(M - 1) * M + a] = make_float2(tm[c], __int_as_float(tk[c]));
        }
    }
    // one tile: the roots sit in a register (lane m holds xi_m) and reach the recursion through v_readlane — an LDS
    // read per step would put its latency on the critical path of this single-wave chain
    const float xv = (tiles == 1 && lane < M) ? Q.lxi[lane] : 0.f;
    for (int m = M - 1; m >= 1; --m) {
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
                T_scratch[(size_t)(m - 1) * M + a] = make_float2(tm[c], __int_as_float(tk[c]));
            }
        }
    }
    __threadfence(); // the rows are read back by the other waves of this workgroup (after its barrier)
}

// The forward sweep.  The recursion P_{m+1} = P_m (1 + xi_m x) is a short dependent chain per step; what is long is the
// dot product D_m = <P_m, T_{m+1}> (two wave reductions and a log) — and nothing depends on it.  So PHD_FW waves of the
// workgroup run the (cheap) recursion redundantly and each takes the dot products of the steps m = wave (mod PHD_FW) only:
// PHD_FW dot products in flight instead of one, no data exchanged between the waves (eight waves are no faster than four:
// the redundant recursion is issue capacity the other resident workgroup can use).
#ifndef PHD_FW
#define PHD_FW 4
#endif
template <int tiles>
__device__ __forceinline__ void cphd_esf_forward(const CphdLds& Q, const float2* __restrict__ T_scratch, int M, int lane, int wave,
                                                 float llam, float lam)
{
#pragma clang fp contract(off)
    if (wave >= PHD_FW) return;   // the recursion is redundant work: only this many waves take part
    const float LOG0F = -FLT_MAX;
    const int XF_ZERO_K = -(1 << 28);
    // P_m[a], a = lane + 1 + 64 c in registers (P_m[0] = 1 is implicit)
    float pm[4] = {0.f, 0.f, 0.f, 0.f};
    int pk[4] = {XF_ZERO_K, XF_ZERO_K, XF_ZERO_K, XF_ZERO_K};
    // this wave's rows come back from L2 / HBM: keep PF of them in flight ahead of the step that uses them
    constexpr int PF = 4;
    float2 rbuf[PF][4], r0buf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int mu = wave + PHD_FW * u;                 // this wave's u-th step
        const float2* row = T_scratch + (size_t)mu * M;
        r0buf[u] = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 4; ++c) rbuf[u][c] = make_float2(0.f, 0.f);
        if (mu < M) {
            r0buf[u] = row[0];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < tiles && lane + 1 + 64 * c <= mu) rbuf[u][c] = row[lane + 1 + 64 * c];
        }
    }
    const float xv = (tiles == 1 && lane < M) ? Q.lxi[lane] : 0.f;   // as in the backward sweep
    for (int m0 = 0; m0 < M; m0 += PF * PHD_FW) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
    for (int r = 0; r < PHD_FW; ++r) {
        const int m = m0 + PHD_FW * u + r;
        if (m < M) {
        const float x = (tiles == 1) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), m)) : Q.lxi[m];
        if (r == wave) {
        // D_m = T_{m+1}[0] + sum_{a=1..m} P_m[a] T_{m+1}[a]
        float qm[5];
        int qk[5];
        int kmax = 2 * XF_ZERO_K;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int a = lane + 1 + 64 * c;
            qm[c] = 0.f; qk[c] = 2 * XF_ZERO_K;
            if (c < tiles && a <= m) {
                const float2 t = rbuf[u][c];
                qm[c] = pm[c] * t.x;
                qk[c] = pk[c] + __float_as_int(t.y);
            }
            kmax = max(kmax, qk[c]);
        }
        {
            const float2 t0 = r0buf[u];
            qm[4] = (lane == 0) ? t0.x : 0.f;
            qk[4] = (lane == 0) ? __float_as_int(t0.y) : 2 * XF_ZERO_K;
            kmax = max(kmax, qk[4]);
        }
        if (m + PF * PHD_FW < M) { // refill this slot with the row of this wave's step PF turns ahead
            const float2* row = T_scratch + (size_t)(m + PF * PHD_FW) * M;
            r0buf[u] = row[0];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < tiles && lane + 1 + 64 * c <= m + PF * PHD_FW) rbuf[u][c] = row[lane + 1 + 64 * c];
        }
        kmax = wave_max_i(kmax);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) s += ldexpf(qm[c], qk[c] - kmax);
        s = wave_sum(s);
        if (lane == 0) {
            int dk = 0;
            const float dm = frexpf(s, &dk);
            Q.lD[m] = dm > 0.f ? logf(dm) + (float)(kmax + dk) * 0.69314718f : LOG0F;   // log <Y1[Z \ m], p>
        }
        } // this wave's step
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
        } // m < M
    }
    }
    }
    if (wave != 0) return;
    // full set: e_j = P_M[j]; <Y0,p> and <Y1,p>
    float ev[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ev[c] = pm[c] > 0.f ? logf(pm[c]) + (float)pk[c] * 0.69314718f : LOG0F;
    float t0[5], t1[5];
    float mx0 = LOG0F, mx1 = LOG0F;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        // c == 4: the j = 0 term (e_0 = 1), carried by lane 0
        const int j = (c == 4) ? 0 : lane + 1 + 64 * c;
        const bool ok = (c == 4) ? (lane == 0) : (c < tiles && j <= M);
        const float e = (c == 4) ? 0.f : ev[c < 4 ? c : 0];
        const float kterm = (float)(M - j) * llam - lam;   // (M-j)! p_K(M-j), Poisson clutter (.bak:398-400)
        t0[c] = ok ? e + Q.I0[ok ? j : 0] + kterm : LOG0F;
        t1[c] = ok ? e + Q.I1[ok ? j : 0] + kterm : LOG0F;
        mx0 = fmaxf(mx0, t0[c]); mx1 = fmaxf(mx1, t1[c]);
    }
    mx0 = wave_max_f(mx0); mx1 = wave_max_f(mx1);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int j = (c == 4) ? 0 : lane + 1 + 64 * c;
        const bool ok = (c == 4) ? (lane == 0) : (c < tiles && j <= M);
        if (ok) { s0 += expf(t0[c] - mx0); s1 += expf(t1[c] - mx1); }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { Q.scal[CQ_LY0] = safe_log(s0) + mx0; Q.scal[CQ_LY1] = safe_log(s1) + mx1; Q.efull[0] = 0.f; }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < tiles && lane + 1 + 64 * c <= M) Q.efull[lane + 1 + 64 * c] = ev[c];
}

template <int CH>
__device__ __forceinline__ void cphd_nsums(const CphdLds& Q, int M, int Nmax, int lane, int wave, float lWq, float lW1)
{
#pragma clang fp contract(off)
    const float LOG0F = -FLT_MAX;
    for (int j = wave; j <= M + 1; j += PHD_NW) {
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
    // cq (diagnostic instantiation, thread 0): time of [staging .. n-sums, backward sweep, forward sweep, rest]
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
    // predicted cardinality (.bak:518-545)
    for (int n = tid; n <= Nmax; n += PHD_T) {
        const int kmax = n < Kb ? n : Kb;
        float mx = Q.cnb[0] + Q.cnq[n];
        for (int k = 1; k <= kmax; ++k) mx = fmaxf(mx, Q.cnb[k] + Q.cnq[n - k]);
        float s = 0.f;
        for (int k = 0; k <= kmax; ++k) s += __expf(Q.cnb[k] + Q.cnq[n - k] - mx);
        Q.cnp[n] = safe_log(s) + mx;
    }
    __syncthreads();
    // I_u[j] = log sum_n p(n) P(n,j+u) Wq^(n-j-u) / W1^n.  Since P(n,j+1) Wq^(n-j-1) is the u = 0 term of j+1,
    // I_1[j] = I_0[j+1] (the same floating-point expression): one family J[j] = I_0[j], j = 0..M+1.
    // Wave per j, lanes over n; the terms stay in registers between the max and the sum pass (n <= 1023).
    // (the chunk count is a compile-time constant per cardinality length: max_cardinality 255 needs 4 of the 16)
    if (cn_len <= 256) cphd_nsums<4>(Q, M, Nmax, lane, wave, lWq, lW1);
    else if (cn_len <= 512) cphd_nsums<8>(Q, M, Nmax, lane, wave, lWq, lW1);
    else cphd_nsums<16>(Q, M, Nmax, lane, wave, lWq, lW1);
    __syncthreads();
    // ESFs (.bak:1224-1272).  The .bak runs one full recursion per left-out measurement (O(M^3)); here
    //   e(Xi \ m) = P_m (*) S_{m+1}   (ESFs of the roots before and after m), so
    //   <Y1[Z\m],p> = sum_a P_m[a] T_{m+1}[a],  T_{m+1}[a] = sum_b S_{m+1}[b] c_{a+b},  c_j = exp(I1[j]) lambda^(M-1-j) e^-lambda
    // and T obeys the same one-root recursion run backwards, T_m[a] = T_{m+1}[a] + xi_m T_{m+1}[a+1], T_M = c:
    // O(M^2), all terms positive.  One wave: a backward sweep that parks the rows T_{m+1}[0..m] in HBM scratch
    // (M^2 x 8 B per particle — what 288 GB are for; they come back out of L2), then a forward sweep that
    // carries P in registers and takes one dot product per measurement.  Values span hundreds of decades, so
    // each is a float mantissa with its own integer exponent (m 2^k): align with v_ldexp, renormalise with
    // v_frexp — exact operations around one correctly rounded multiply and add (the oracle does the same).
    const int tiles = (M + 63) >> 6;
    CQSTAMP(1);
    if (wave == 0) {
        if (tiles == 1) cphd_esf_backward<1>(Q, T_scratch, M, lane, llam, lam);
        else if (tiles == 2) cphd_esf_backward<2>(Q, T_scratch, M, lane, llam, lam);
        else cphd_esf_backward<4>(Q, T_scratch, M, lane, llam, lam);
    }
    __syncthreads();
    CQSTAMP(2);
    if (tiles == 1) cphd_esf_forward<1>(Q, T_scratch, M, lane, wave, llam, lam);
    else if (tiles == 2) cphd_esf_forward<2>(Q, T_scratch, M, lane, wave, llam, lam);
    else cphd_esf_forward<4>(Q, T_scratch, M, lane, wave, llam, lam);
    __syncthreads();
    CQSTAMP(3);
    const float lY0 = Q.scal[CQ_LY0];
    for (int m = tid; m < M; m += PHD_T) L.logZ[m] = -((llam - lkap) + Q.lD[m] - lY0);      // .bak:1434-1437
    if (tid == 0) Q.scal[CQ_R1] = expf(Q.scal[CQ_LY1] - lY0);                               // .bak:1452-1455
    // updated cardinality (.bak:1409-1411)
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

// (defined further down) the weights / nEff / resample routine, run by the last workgroup of a fused step
template <int BT, int R, bool HANDOFF>
__device__ __forceinline__ void weights_body(const WeightArgs& A, unsigned char* s_dyn);

// ------------------------------------------------------------------------------------------
// the fused update + prune + merge kernel
// ------------------------------------------------------------------------------------------
template <bool STAMPS, bool FUSEW, bool CPHD>
__global__ __launch_bounds__(PHD_T, PHD_MIN_WAVES) void phd_update_merge_kernel(UpdateArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const Lds L = lds_carve((lds_u8)lds_raw, A.S_cap, A.cap, A.MM);
    // CPHD instantiation: its arrays follow the common layout
    const CphdLds Q = CPHD ? cphd_carve((lds_u8)lds_raw + lds_offsets(A.S_cap, A.cap, A.MM).total, A.cn_len, A.MM) : CphdLds();

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const int p = blockIdx.x;
    const DevConfig& cfg = A.cfg;
    const int cap = A.cap, S_cap = A.S_cap, M = A.M;
    const int src = A.parent[p];
    const float* __restrict__ in = A.map_in + (size_t)src * 6 * cap;
    const unsigned rows_stride = FUSEW ? 0u : A.out_stride; // rows mode belongs to the multi-GPU step (never the fused tail)
    float* __restrict__ out = A.map_out + (size_t)p * (rows_stride ? rows_stride : (size_t)6 * cap);
    const int n_map = A.count_in[src];
    phd_pose pose = A.pose[p];
    if (A.do_predict) {
        // fused vehicle predict: every lane computes the same pose (no broadcast needed), lane 0 stores it
        float n_alpha, n_encoder;
        if (A.noise) { n_alpha = A.noise[p].n_alpha; n_encoder = A.noise[p].n_encoder; }
        else draw_noise(A.seed, A.counter, p, cfg, n_alpha, n_encoder);
        pose = predict_pose(pose, A.control, n_alpha, n_encoder, cfg);
        if (tid == 0) {
            if (FUSEW) { // handed to the last workgroup: agent-scope (sc1) stores
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
    }
    u64* st = STAMPS ? (A.stamps + (size_t)p * 16) : nullptr;
    u64 cq[5] = {0, 0, 0, 0, 0};
    if (STAMPS && tid == 0) { st[12] = 0; st[13] = 0; st[14] = 0; st[15] = 0; }
    STAMP(0);

    if (tid < 32) L.ctr[tid] = 0;
    // measurements -> LDS (the reference keeps them in __constant__ Z[256], src/phdfilter.cu:120)
    for (int m = tid; m < M; m += PHD_T) {
        phd_measurement z = A.z[m];
        L.z_r[m] = z.range;
        L.z_b[m] = z.bearing;
        L.zok[m] = (z.label == 0 || !cfg.labeledMeasurements) ? 1u : 0u; // :1913
    }
    __syncthreads();

    // ---- birth geometry (host loop src/phdfilter.cu:3470-3506): depends only on the pose and the
    //      measurement, so the top lanes of the workgroup — idle while the low lanes classify the map —
    //      compute it now; the birth weights follow once the normalisers are known
    for (int m = PHD_T - 1 - tid; m < M; m += PHD_T) {
        const float zr = L.z_r[m];
        const float theta = pose.ptheta + L.z_b[m];
        float sn, cs;
        sincosf(theta, &sn, &cs);
        const float dx = zr * cs, dy = zr * sn;
        const float J0 = dx / zr, J1 = dy / zr, J2 = -dy, J3 = dx;
        const float sr = cfg.stdRange * cfg.birthNoiseFactor, sb = cfg.stdBearing * cfg.birthNoiseFactor;
        const float vr = sr * sr, vb = sb * sb;
        L.bgeo[0 * A.MM + m] = pose.px + dx;
        L.bgeo[1 * A.MM + m] = pose.py + dy;
        L.bgeo[2 * A.MM + m] = J0 * J0 * vr + J2 * J2 * vb;
        L.bgeo[3 * A.MM + m] = J0 * J1 * vr + J2 * J3 * vb;
        L.bgeo[4 * A.MM + m] = J1 * J1 * vr + J3 * J3 * vb;
    }

    // ---- classification + per-feature EKF terms -----------------------------------------------
    float pdw_local = 0.f; // sum_j pd_j w_j (cardinality_predict, :2160)
    float wall_local = 0.f; // CPHD: <1, map>
    {
        int n_in = 0, n_out0 = 0;
        for (int i0 = 0; i0 < n_map; i0 += PHD_T) {
            const int i = i0 + tid;
            int cls = -1, nd_j = 0;
            float w = 0, mx = 0, my = 0, pxx = 0, pxy = 0, pyy = 0;
            EkfTerms t;
            t.pd = 0.f;
            if (i < n_map) {
                w = in[0 * cap + i]; mx = in[1 * cap + i]; my = in[2 * cap + i];
                pxx = in[3 * cap + i]; pxy = in[4 * cap + i]; pyy = in[5 * cap + i];
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
                {
                    // gain and Joseph covariance do not depend on the measurement (src/phdfilter.cu:1884-1901):
                    // once per feature here, not once per surviving detection term in the finalise pass
                    float oxx, oxy, oyy;
                    joseph_cov(t, pxx, pxy, pyy, cfg, oxx, oxy, oyy);
                    L.f_k[j] = (v4f){t.K0, t.K1, t.K2, t.K3};
                    L.f_p[j] = (v4f){oxx, oxy, oyy, mx};
                    L.f_my[j] = my;
                }
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
                                         cls == 2 ? NEAR_U_BASE + i : nd_j);
            }
            n_in += tot_in; n_out0 += tot_out;
            __syncthreads();
        }
        if (tid == 0) { L.ctr[CTR_NIN] = n_in; L.ctr[CTR_NOUT] = n_out0; }
    }
    __syncthreads();
    const int n_in = L.ctr[CTR_NIN];
    const int n_out0 = L.ctr[CTR_NOUT];
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
    int ncw = 0; // terms listed by this wave
    // ---- pass 1: normalisers ----------------------------------------------------------------------
    for (int mt = 0; mt < m_tiles; ++mt) {
        const int m = mt * 64 + lm;
        const bool mvalid = (m < M) && L.zok[m < M ? m : 0];
        const float zr = L.z_r[m < M ? m : 0], zb = L.z_b[m < M ? m : 0];
        float acc = 0.f;
        for (int jb = wave * JS; jb < n_in; jb += PHD_NW * JS) {
            const int j = jb + js;
            const int jj = j < n_in ? j : n_in - 1;
            const v4f fa = L.f_a[jj];
            const v2f fc = L.f_c[jj];
            const float i0 = zr - fa.x;
            const float i1 = wrap_angle(zb - fa.y);
            const float dist = i0 * i0 * fa.z + i0 * i1 * fa.w + i1 * i1 * fc.x;                       // :1908-1910
            const float lw = fc.y - 0.5f * dist;
            const float e = __expf(lw);                                                               // :2205
            acc += (mvalid && j < n_in) ? e : 0.f;
            if (sparse2) {
                // every wave fills its own segment of the list: positions from a ballot, no atomics in the loop
                const bool cnd = mvalid && (j < n_in) && !(lw < c0m);   // NaN stays a candidate, as in the dense test
                const u64 bal = __ballot(cnd);
                const int pos = ncw + __popcll(bal & lanemask_lt());
                if (cnd && pos < PHD_CAND_SEG) clist[wave * PHD_CAND_SEG + pos] = (u16)(m * n_in + j);
                ncw += __popcll(bal);
            }
        }
        for (int off = Mp; off < 64; off <<= 1) acc += xor_lane(acc, off);
        if (js == 0 && m < M) L.zpart[wave * A.MM + m] = acc;
    }
    if (sparse2 && lane == 0) L.ctr[CTR_TMP + wave] = ncw; // the per-wave slots are free between the classification and the merge
    __syncthreads();
    float lz_local = 0.f;
    if (CPHD) {
        // roots Xi_m = (lambda/kappa)(sum_j pd w_j g_jm + birthWeight) (.bak:1205-1222)
        const float rat = cfg.clutterRate / cfg.clutterDensity;
        for (int m = tid; m < M; m += PHD_T) {
            float sum = L.zpart[0 * A.MM + m];
#pragma unroll
            for (int wv = 1; wv < PHD_NW; ++wv) sum += L.zpart[wv * A.MM + m];
            Q.lxi[m] = (sum + cfg.birthWeight) * rat;
        }
        const float pdw = block_sum(pdw_local, L.red, tid);
        const float w_all = block_sum(wall_local, L.red, tid);
        __syncthreads();
        cphd_block(L, Q, cfg, M, A.MM, A.cn_len, A.lfact, A.lfact_len, A.cn_in + (size_t)src * A.cn_len,
                   A.cn_out + (size_t)p * (rows_stride ? rows_stride : (unsigned)A.cn_len), A.cphd_scratch + (size_t)p * A.MM * A.MM, w_all, pdw, tid,
                   STAMPS ? cq : nullptr);
        const float r1 = Q.scal[CQ_R1];
        // births (weight bw (lambda/kappa) <Y1[Z\m],p>/<Y0,p>)
        for (int m0 = 0; m0 < M; m0 += PHD_T) {
            const int m = m0 + tid;
            const bool mv = m < M;
            const float wb = (mv && L.zok[mv ? m : 0]) ? expf(safe_log(cfg.birthWeight) - L.logZ[mv ? m : 0]) : 0.f;
            const bool keep = mv && !(wb < cfg.minFeatureWeight);
            const int slot = alloc_slots(keep, L.ctr);
            if (keep) store_survivor(L, slot, S_cap, wb, L.bgeo[0 * A.MM + m], L.bgeo[1 * A.MM + m], L.bgeo[2 * A.MM + m],
                                     L.bgeo[3 * A.MM + m], L.bgeo[4 * A.MM + m], n_in + M * n_in + m);
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
                                     in[5 * cap + i], j);
        }
        // nearly-in-range features (pD = 0): weight w r1, join the merge unpruned like HEAD (:3242-3257)
        const int n_near = L.ctr[CTR_NNEAR];
        for (int k0 = 0; k0 < n_near; k0 += PHD_T) {
            const int k = k0 + tid;
            const bool kv = k < n_near;
            const int i = L.out_idx[cap - 1 - (kv ? k : 0)];
            const int slot = alloc_slots(kv, L.ctr);
            if (kv) store_survivor(L, slot, S_cap, in[0 * cap + i] * r1, in[1 * cap + i], in[2 * cap + i], in[3 * cap + i],
                                   in[4 * cap + i], in[5 * cap + i], NEAR_U_BASE + i);
        }
        if (tid == 0) {
            A.dlogw[p] = Q.scal[CQ_LY0];                                                             // .bak:2661-2667
            if (A.raw_out) A.raw_out[p] = A.logw_in[p] + Q.scal[CQ_LY0];
        }
    } else {
    for (int m = tid; m < M; m += PHD_T) {
        float sum = L.zpart[0 * A.MM + m];
#pragma unroll
        for (int wv = 1; wv < PHD_NW; ++wv) sum += L.zpart[wv * A.MM + m];
        sum += cfg.clutterDensity;                                                                    // :2213
        sum += cfg.birthWeight;                                                                       // :2214
        const float lz = safe_log(sum);                                                               // :2217
        L.logZ[m] = lz;
        lz_local += lz;                                                                               // :2251
        // the birth term of this measurement (weight :2239-2243, prune :2314) while lz is in a register
        const float wb = L.zok[m] ? expf(safe_log(cfg.birthWeight) - lz) : 0.f;
        const bool keep = !(wb < cfg.minFeatureWeight);
        const int slot = alloc_slots(keep, L.ctr);
        if (keep) store_survivor(L, slot, S_cap, wb, L.bgeo[0 * A.MM + m], L.bgeo[1 * A.MM + m], L.bgeo[2 * A.MM + m],
                                 L.bgeo[3 * A.MM + m], L.bgeo[4 * A.MM + m], n_in + M * n_in + m);
    }
    {
        float lz_sum, pdw;
        block_sum2(lz_local, pdw_local, L.red, tid, lz_sum, pdw);
        // particle_weighting == 0 (:2260-2263): sum_m log Z_m - (sum_j pd_j w_j + M * birthWeight)
        if (tid == 0) {
            const float dl = lz_sum - (pdw + (float)M * cfg.birthWeight);
            if (FUSEW) __hip_atomic_store(&A.dlogw[p], dl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else A.dlogw[p] = dl;
            if (!FUSEW && A.raw_out) A.raw_out[p] = A.logw_in[p] + dl;                                // :3741-3744
        }
    }
    } // !CPHD
    __syncthreads();

    STAMP(2);
    // ---- pass 2: final weights; prune before store --------------------------------------------------
    STAMP(3);
    // detection terms
    int n_cand = 0, seg_max = 0;
    if (sparse2) {
#pragma unroll
        for (int wv = 0; wv < PHD_NW; ++wv) {
            const int c = L.ctr[CTR_TMP + wv];
            n_cand += c;
            seg_max = c > seg_max ? c : seg_max;
        }
    }
    if (sparse2 && seg_max <= PHD_CAND_SEG) {
        const float rcp_nin = 1.0f / (float)(n_in > 0 ? n_in : 1);
        for (int t0 = 0; t0 < n_cand; t0 += PHD_T) {
            const int t = t0 + tid;
            const bool tv = t < n_cand;
            // entry t of the concatenated segments
            int sw = 0, so = tv ? t : 0;
#pragma unroll
            for (int wv = 0; wv < PHD_NW - 1; ++wv) {
                const int c = L.ctr[CTR_TMP + wv];
                const bool next = (sw == wv) && (so >= c);
                so = next ? so - c : so;
                sw = next ? wv + 1 : sw;
            }
            const int idx = tv ? clist[sw * PHD_CAND_SEG + so] : 0;
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
                if (slot < S_cap) { L.w[slot] = w; L.u[slot] = n_in + m * n_in + j; }
                else L.ctr[CTR_OVERFLOW] = 1;
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
                if (slot < S_cap) { L.w[slot] = w; L.u[slot] = n_in + m * n_in + j; }
                else L.ctr[CTR_OVERFLOW] = 1;
            }
        }
    }
    STAMP(4);
    __syncthreads();
    {
        // dense finalise of the detection terms: rebuild gain, updated mean and Joseph covariance
        // from the prior feature (src/phdfilter.cu:1884-1906)
        const int n_now = min(L.ctr[CTR_NSURV], S_cap);
        const int u_det_lo = n_in, u_det_hi = n_in + M * n_in;
        for (int s = tid; s < n_now; s += PHD_T) {
            const int u = L.u[s];
            if (u < u_det_lo || u >= u_det_hi) continue;
            const int m = (u - n_in) / n_in;
            const int j = (u - n_in) - m * n_in;
            const v4f fa = L.f_a[j], K = L.f_k[j], Pn = L.f_p[j];
            const float i0 = L.z_r[m] - fa.x;
            const float i1 = wrap_angle(L.z_b[m] - fa.y);
            L.mx[s] = Pn.w + K.x * i0 + K.z * i1;                                                    // :1903-1904
            L.my[s] = L.f_my[j] + K.y * i0 + K.w * i1;
            L.xx[s] = Pn.x; L.xy[s] = Pn.y; L.yy[s] = Pn.z;
        }
    }
    __syncthreads();
    STAMP(5);
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
    const int n_update = n_in * (M + 1) + M;
    const bool packed = (n_update + n_map <= 0xFFFF) && (S_cap <= 0x10000);
    if (cfg.distanceMetric == 0) merge_in_lds<false, STAMPS>(L, S_cap, n_surv, cfg, out, cap, tid, st, n_update, packed);
    else merge_in_lds<true, STAMPS>(L, S_cap, n_surv, cfg, out, cap, tid, st, n_update, packed);
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
        if (A.parent_reset) { // the output slab of particle p is its own again
            if (FUSEW) __hip_atomic_store(&A.parent_reset[p], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else A.parent_reset[p] = p;
        }
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
        atomicMax(A.max_surv, L.ctr[CTR_NSURV]);
        atomicMax(A.max_map, k_out + n_out0);
    }
    STAMP(11);
    if (STAMPS && CPHD && tid == 0) { // the CPHD block's parts replace the merge-round statistics
        st[12] = cq[1] - cq[0]; st[13] = cq[2] - cq[1]; st[14] = cq[3] - cq[2]; st[15] = cq[4] - cq[3];
    }
    if (FUSEW) {
        // ---- fused tail: the workgroup that finishes last runs the weights / nEff / resample routine.
        // Hand-off (cdna guide, Guideline 16): lane 0 of every workgroup made its hand-off stores
        // (dlogw, predicted pose, parent reset) agent-scope, drains them, then takes a ticket with an
        // agent-scope atomic; the workgroup whose ticket is the last one reads them with sc1 loads.
        lds_i32 flag = L.ctr + CTR_TMP;
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == gridDim.x - 1);
            *flag = last;
            if (last) __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
        }
        __syncthreads();
        if (*flag) weights_body<PHD_T, 2, true>(A.wa, lds_raw);
    }
}

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

// ------------------------------------------------------------------------------------------
// particle weights: accumulate, logSumExp normalise, nEff, resample (one workgroup)
// ------------------------------------------------------------------------------------------
// portable exp for the resampling CDF: IEEE basic operations only (mul, fma, rint, ldexp), so
// the double it returns is the same on every conforming CPU and GPU (see oracle/scphd_cpu.c).
// -> det_exp() in phd_detexp.h (shared with phd_eap.hip)

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
// the total (carry included).  Thread t owns the contiguous entries [t*per, (t+1)*per).
template <int BT>
__device__ __forceinline__ u64 block_scan_u64(u64* q, int m, u64 carry, u64* s_wtot, int tid)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const int per = (m + BT - 1) / BT;
    const int lo = tid * per, hi = (lo + per < m) ? lo + per : m;
    u64 local = 0;
    for (int e = lo; e < hi; ++e) local += q[e];
    const u64 incl = wave_incl_scan(local);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    u64 woff = carry, total = carry;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) {
        const u64 c = s_wtot[w];
        if (w < wave) woff += c;
        total += c;
    }
    u64 run = woff + incl - local;
    for (int e = lo; e < hi; ++e) { run += q[e]; q[e] = run; }
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

template <int BT, int R, bool HANDOFF>
__device__ __forceinline__ void weights_body(const WeightArgs& A, unsigned char* s_dyn)
{
    __shared__ float sc[BT / 64];
    __shared__ int s_argmax;
    __shared__ u64 s_wtot[BT / 64];
    __shared__ double s_bestv[BT / 64];
    __shared__ int s_besti[BT / 64];
    const int tid = threadIdx.x;
    const int n = A.n;
#define WSTAMP(k) do { if (A.wstamps && tid == 0) A.wstamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
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
            if (A.raw_out) A.raw_out[i] = w[r];
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
    if (tid == 0) { A.neff_out[0] = neff; A.did_resample[0] = doit; }
    const int n_new = A.n_new;
    if (!doit) { // uniform: neff is the same in every thread
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * BT;
            if (i < n) A.logw[i] = w[r];
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
    for (int j = tid; j < n_new; j += BT) {
        const double u = (A.n_uniforms == 1) ? A.u0 : A.uniforms[j];
        const double r = j * interval + u * interval;                                                  // :468
        const u64 T = (u64)ceil(r * scale);
        int idx;
        if (T > ctot) {
            idx = s_argmax;
        } else {
            int lo = 0, hi = n - 1;
            while (lo < hi) { // smallest i with Q_i >= T
                const int mid = (lo + hi) >> 1;
                if (Q[mid] < T) lo = mid + 1; else hi = mid;
            }
            idx = lo;
        }
        A.idx_out[j] = idx;
        if (A.mode & W_COMMIT) { // copy_particles (src/slamtypes.h:313-333)
            A.pose_out[j] = ld_pose<HANDOFF>(&A.pose_in[idx]);
            A.parent_out[j] = ld_i32<HANDOFF>(&A.parent_in[idx]);
        }
    }
    // weights: -log(N) after a committed resample (slamtypes.h:327), else the normalised values
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * BT;
        if (i < n) A.logw[i] = (A.mode & W_COMMIT) ? nlw : w[r];
    }
    WSTAMP(6);
}

template <int BT, int R>
__global__ __launch_bounds__(BT) void phd_weights_small_kernel(WeightArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_wdyn[];
    weights_body<BT, R, false>(A, s_wdyn);
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
hipError_t launch_update_merge(const UpdateArgs& a, int n_particles, size_t lds_bytes, hipStream_t st)
{
    // function attributes are per device: set once for every device this process launches on
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    if (!attr_set) {
        // dynamic LDS up to the CU's 160 KiB minus what the instantiation declares statically
        const void* fns[5] = {(const void*)phd_update_merge_kernel<false, false, false>,
                              (const void*)phd_update_merge_kernel<true, false, false>,
                              (const void*)phd_update_merge_kernel<false, true, false>,
                              (const void*)phd_update_merge_kernel<false, false, true>,
                              (const void*)phd_update_merge_kernel<true, false, true>};
        for (int k = 0; k < 5; ++k) {
            hipFuncAttributes fa;
            hipError_t e = hipFuncGetAttributes(&fa, fns[k]);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    if (a.cphd && a.stamps) hipLaunchKernelGGL((phd_update_merge_kernel<true, false, true>), dim3(n_particles), dim3(PHD_T), lds_bytes, st, a);
    else if (a.cphd) hipLaunchKernelGGL((phd_update_merge_kernel<false, false, true>), dim3(n_particles), dim3(PHD_T), lds_bytes, st, a);
    else if (a.fuse_weights) hipLaunchKernelGGL((phd_update_merge_kernel<false, true, false>), dim3(n_particles), dim3(PHD_T), lds_bytes, st, a);
    else if (a.stamps) hipLaunchKernelGGL((phd_update_merge_kernel<true, false, false>), dim3(n_particles), dim3(PHD_T), lds_bytes, st, a);
    else hipLaunchKernelGGL((phd_update_merge_kernel<false, false, false>), dim3(n_particles), dim3(PHD_T), lds_bytes, st, a);
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

hipError_t launch_weights(const WeightArgs& a, hipStream_t st)
{
    // one workgroup; small particle sets use a small one (cheaper barriers, same results: the
    // reductions are fixed trees per block size)
    // (the commit of the small kernel writes logw[j] for j < n: it needs n_new <= n, true for every caller)
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)phd_weights_small_kernel<1024, 16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t dyn = (size_t)a.n * 8; // the fixed-point CDF, one u64 per particle
    if (a.n <= 512 && a.n_new <= a.n) hipLaunchKernelGGL((phd_weights_small_kernel<256, 2>), dim3(1), dim3(256), dyn, st, a);
    else if (a.n <= 2048 && a.n_new <= a.n) hipLaunchKernelGGL((phd_weights_small_kernel<256, 8>), dim3(1), dim3(256), dyn, st, a);
    else if (a.n <= 16384 && a.n_new <= a.n) hipLaunchKernelGGL((phd_weights_small_kernel<1024, 16>), dim3(1), dim3(1024), dyn, st, a);
    else if (a.n <= 1024) hipLaunchKernelGGL(phd_weights_kernel<256>, dim3(1), dim3(256), 0, st, a);
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

} // namespace phd
