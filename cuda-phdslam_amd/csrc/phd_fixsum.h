// phd_fixsum.h — exact, order-free moment sums of the merge (device code, namespace phd).
//
// The reference adds a cluster's weights, weighted means and weighted covariances with a block-size-dependent reduction
// tree (src/phdfilter.cu:2795-2881, sumByReduction), i.e. in no particular order.  A floating-point sum depends on that
// order; an integer sum does not.  Every term the reference adds is therefore converted — exactly, by shifting its
// mantissa — to a fixed-point integer on a scale anchored at the cluster's own magnitudes, the integers are added (LDS
// atomics in any order, or a sequential loop: same result), and the three divisions of the reference's formulas
// (:2828, :2879) are taken once, in double.  The oracle (oracle/scphd_cpu.c, o_merge_exact) restates the same arithmetic
// independently; device and oracle agree bit for bit.
//
//   term                         float value t (the reference's expression)           anchor field F     bits
//   weight                       w_i                                                  Fw = field(w_seed)  16 -> q < 2^40
//   weighted mean (x, y)         w_i * m_i                      (:2813)               Fw + 28             38 -> q < 2^62, two 31-bit digits
//   weighted covariance (3)      w_i * (P_i + d d^T), d = mean - m_i   (:2854-2866)   max(1, Fw + ec - 121)  18 -> q < 2^42
//
// field(x) = the biased exponent of x (1 for denormals and zero): |x| < 2^(field - 126).  The seed is the heaviest member, so
// every weight fits below Fw.  ec = the largest field among the members' covariance entries and squared offsets from the
// seed's mean: |P + d d^T| < 2^(ec - 123.6) because the merged mean lies within twice the largest offset.  A term whose field
// exceeds its anchor (negative weights of larger magnitude than the seed's, coordinates beyond 2^28) or that is not finite
// poisons the cluster: its output is NaN.  Sums of up to 2048 terms stay below 2^53, so their conversion to double is exact.
#pragma once
#include "phd_defs.h"

namespace phd {

typedef long long i64;

struct FxFields {
    u32 m24;    // mantissa with the implicit bit (denormals: without)
    int e1;     // biased exponent, 1 for denormals / zero: value = +-m24 * 2^(e1 - 150)
    bool neg, bad;
};

__device__ __forceinline__ FxFields fx_fields(float x)
{
    const u32 b = __float_as_uint(x);
    const u32 ef = (b >> 23) & 0xFFu, frac = b & 0x7FFFFFu;
    FxFields f;
    f.neg = (b >> 31) != 0u;
    f.bad = ef == 255u;
    f.m24 = ef ? (frac | 0x800000u) : frac;
    f.e1 = ef ? (int)ef : 1;
    return f;
}

// the field alone, 255 for a non-finite value
__device__ __forceinline__ int fx_field(float x)
{
    const u32 ef = (__float_as_uint(x) >> 23) & 0xFFu;
    return ef ? (int)ef : 1;
}

// q = trunc(|x| * 2^(top + 150 - F)) with x's sign: x ~ q * 2^(F - top - 150).  ok = finite and field(x) <= F.
__device__ __forceinline__ i64 fx_q(float x, int F, int top, bool& ok)
{
    const FxFields f = fx_fields(x);
    ok = ok && !f.bad && f.e1 <= F;
    const int k = f.e1 - F + top;
    u64 q = 0ull;
    if (k >= 0) q = (u64)f.m24 << (k > 40 ? 40 : k);      // (k <= top when ok; the clamp only keeps the shift defined)
    else if (k > -32) q = (u64)(f.m24 >> (-k));
    return f.neg ? -(i64)q : (i64)q;
}

// the two 31-bit digits of a 62-bit term: q = hi * 2^31 + lo, both carrying the sign
__device__ __forceinline__ void fx_digits(i64 q, i64& hi, i64& lo)
{
    const bool neg = q < 0;
    const u64 a = neg ? (u64)(-q) : (u64)q;
    const i64 h = (i64)(a >> 31), l = (i64)(a & 0x7FFFFFFFull);
    hi = neg ? -h : h;
    lo = neg ? -l : l;
}

#define FX_W_TOP 16
#define FX_M_TOP 38
#define FX_M_HEAD 28
#define FX_C_TOP 18

__device__ __forceinline__ int fx_cov_anchor(int Fw, int ec)
{
    const int F = Fw + ec - 121;
    return F < 1 ? 1 : F;
}

// one cluster's sums.  Sequential users (merge_small, the spill merge) keep one in registers; the round-based merge keeps
// them in LDS and adds with 64-bit atomics — integer addition commutes, the results are the same.
struct FxSums {
    i64 W, xh, xl, yh, yl;
    i64 cxx, cxy, cyy;
    int ec;
    bool ok;
};

// pass 1 term of member (w, mx, my, P) of the cluster seeded at (smx, smy) with weight field Fw
__device__ __forceinline__ void fx_add_first(FxSums& s, int Fw, float smx, float smy, float w, float mx, float my, float xx,
                                             float xy, float yy)
{
#pragma clang fp contract(off)
    bool ok = s.ok;
    s.W += fx_q(w, Fw, FX_W_TOP, ok);
    i64 h, l;
    fx_digits(fx_q(w * mx, Fw + FX_M_HEAD, FX_M_TOP, ok), h, l);                  // src/phdfilter.cu:2813
    s.xh += h; s.xl += l;
    fx_digits(fx_q(w * my, Fw + FX_M_HEAD, FX_M_TOP, ok), h, l);
    s.yh += h; s.yl += l;
    const float dx = mx - smx, dy = my - smy;
    int e = fx_field(xx);
    const int e2 = fx_field(xy), e3 = fx_field(yy), e4 = fx_field(dx * dx), e5 = fx_field(dy * dy);
    e = e2 > e ? e2 : e; e = e3 > e ? e3 : e; e = e4 > e ? e4 : e; e = e5 > e ? e5 : e;
    s.ec = e > s.ec ? e : s.ec;
    s.ok = ok;
}

// (:2828) mean = sum(w m) / sum(w), one division in double; W as float
__device__ __forceinline__ void fx_mean(const FxSums& s, int Fw, float& W, float& mx, float& my)
{
    const double Wd = (double)s.W;
    W = (float)__builtin_ldexp(Wd, Fw - 150 - FX_W_TOP);
    const double sx = __builtin_fma((double)s.xh, 2147483648.0, (double)s.xl);
    const double sy = __builtin_fma((double)s.yh, 2147483648.0, (double)s.yl);
    // scale of the mean digits over the scale of the weights: 2^(Fw + 28 - 38 - 150) / 2^(Fw - 16 - 150) = 2^6
    mx = (float)(sx / Wd * 64.0);
    my = (float)(sy / Wd * 64.0);
}

// pass 2 term (:2854-2866): d = merged mean - member mean
__device__ __forceinline__ void fx_cov_terms(int Fc, float mean_x, float mean_y, float w, float mx, float my, float xx, float xy,
                                             float yy, i64& qxx, i64& qxy, i64& qyy, bool& ok)
{
#pragma clang fp contract(off)
    const float d0 = mean_x - mx, d1 = mean_y - my;
    qxx = fx_q(w * (xx + d0 * d0), Fc, FX_C_TOP, ok);
    qxy = fx_q(w * (xy + d0 * d1), Fc, FX_C_TOP, ok);
    qyy = fx_q(w * (yy + d1 * d1), Fc, FX_C_TOP, ok);
}

// (:2879) cov = sum / W
__device__ __forceinline__ float fx_cov(i64 c, i64 Wq, int Fc, int Fw)
{
    return (float)__builtin_ldexp((double)c / (double)Wq, (Fc - FX_C_TOP) - (Fw - FX_W_TOP));
}

// ---- the same integers carried in doubles (the LDS merges) ----------------------------------------------------------------
// A term has 24 significant bits and, when it is `ok`, fewer than 2^53 of magnitude; sums of up to 2048 of them stay below
// 2^53 (see above), so adding them as DOUBLES is exact — still integer arithmetic, still order-free, bit for bit the sums of
// the 64-bit integer path — and the conversion is three instructions (v_cvt_f64_f32, v_ldexp_f64, v_trunc_f64) instead of
// the mantissa surgery of fx_q(); LDS adds doubles atomically (ds_add_f64).  The spill merge (HBM accumulators) keeps integers.
__device__ __forceinline__ double fx_qd(float x, int F, int top, bool& ok)
{
    const u32 ef = (__float_as_uint(x) >> 23) & 0xFFu;
    ok = ok && ef != 255u && (int)(ef ? ef : 1u) <= F;
    return __builtin_trunc(__builtin_ldexp((double)x, top + 150 - F));   // = fx_q(x, F, top): trunc(|x| 2^(top + 150 - F)), signed
}
// the two 31-bit digits (both carrying the sign, like fx_digits): exact, q has 24 significant bits
__device__ __forceinline__ void fx_digits_d(double q, double& hi, double& lo)
{
    hi = __builtin_trunc(q * 4.656612873077392578125e-10);               // 2^-31
    lo = __builtin_fma(hi, -2147483648.0, q);
}
// ... and turned into 64-bit integers for the LDS atomics (round 4): ds_add_u64 retires ~1.7 x the lanes per cycle of ds_add_f64
// (tools/probes/lds_atomic_probe.hip: 3.9 against 2.4 per CU), and the moment sums are bound by exactly that rate.  There is no
// f64 -> i64 convert instruction; for an integer-valued double below 2^50 the sum v + 1.5 * 2^52 is exact and carries v's two's
// complement in its low 51 mantissa bits: one v_add_f64 and one v_bfe_i32 (sign extension of the high word's 19 bits).
__device__ __forceinline__ i64 fx_i64(double v)
{
    const u64 b = (u64)__double_as_longlong(v + 6755399441055744.0);
    const int hi = __builtin_amdgcn_sbfe((int)(b >> 32), 0, 19);
    return (i64)(((u64)(u32)hi << 32) | (u64)(u32)b);
}

struct FxSumsD {
    double W, xh, xl, yh, yl;
    int ec;
    bool ok;
};
// pass 1 term (fx_add_first)
__device__ __forceinline__ void fx_first_d(FxSumsD& s, int Fw, float smx, float smy, float w, float mx, float my, float xx, float xy,
                                           float yy)
{
#pragma clang fp contract(off)
    bool ok = true;
    s.W = fx_qd(w, Fw, FX_W_TOP, ok);
    fx_digits_d(fx_qd(w * mx, Fw + FX_M_HEAD, FX_M_TOP, ok), s.xh, s.xl);         // src/phdfilter.cu:2813
    fx_digits_d(fx_qd(w * my, Fw + FX_M_HEAD, FX_M_TOP, ok), s.yh, s.yl);
    const float dx = mx - smx, dy = my - smy;
    int e = fx_field(xx);
    const int e2 = fx_field(xy), e3 = fx_field(yy), e4 = fx_field(dx * dx), e5 = fx_field(dy * dy);
    e = e2 > e ? e2 : e; e = e3 > e ? e3 : e; e = e4 > e ? e4 : e; e = e5 > e ? e5 : e;
    s.ec = e;
    s.ok = ok;
}
// (:2828) fx_mean on double sums
__device__ __forceinline__ void fx_mean_d(const FxSumsD& s, int Fw, float& W, float& mx, float& my)
{
    W = (float)__builtin_ldexp(s.W, Fw - 150 - FX_W_TOP);
    const double sx = __builtin_fma(s.xh, 2147483648.0, s.xl);
    const double sy = __builtin_fma(s.yh, 2147483648.0, s.yl);
    mx = (float)(sx / s.W * 64.0);
    my = (float)(sy / s.W * 64.0);
}
// pass 2 term (fx_cov_terms)
__device__ __forceinline__ void fx_cov_terms_d(int Fc, float mean_x, float mean_y, float w, float mx, float my, float xx, float xy,
                                               float yy, double& qxx, double& qxy, double& qyy, bool& ok)
{
#pragma clang fp contract(off)
    const float d0 = mean_x - mx, d1 = mean_y - my;
    qxx = fx_qd(w * (xx + d0 * d0), Fc, FX_C_TOP, ok);
    qxy = fx_qd(w * (xy + d0 * d1), Fc, FX_C_TOP, ok);
    qyy = fx_qd(w * (yy + d1 * d1), Fc, FX_C_TOP, ok);
}
// (:2879) fx_cov on double sums
__device__ __forceinline__ float fx_cov_d(double c, double Wq, int Fc, int Fw)
{
    return (float)__builtin_ldexp(c / Wq, (Fc - FX_C_TOP) - (Fw - FX_W_TOP));
}

} // namespace phd
