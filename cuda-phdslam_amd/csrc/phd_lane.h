// phd_lane.h — cross-lane exchange (DPP / permlane), wave and workgroup reductions and scans, small math helpers.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"

namespace phd {

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
// scratch_idle: nobody can still be reading the scratch from an earlier reduction (saves the leading barrier)
__device__ __forceinline__ void block_sum2(float a, float b, LDS_T(float)* scratch, int tid, float& ra, float& rb,
                                           bool scratch_idle = false)
{
    a = wave_sum(a);
    b = wave_sum(b);
    if (!scratch_idle) __syncthreads();
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

} // namespace phd
