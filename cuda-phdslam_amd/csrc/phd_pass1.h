// phd_pass1.h — pass 1 of the PHD update: per-measurement normalisers and the list of terms that can survive the prune.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"

namespace phd {

// wrap_angle() for |a| < 2 pi, branch-free (same operations, same bits)
__device__ __forceinline__ float wrap_angle_small(float a)
{
    const float TWO_PI_F = 6.2831855f;
    const float PI_F = 3.14159274f;
    const float B = -1.7484555e-7f; // 2*pi - float(2*pi)
    const float hi = (a - TWO_PI_F) - B, lo = (a + TWO_PI_F) + B;
    float r = (a <= -PI_F) ? lo : a;
    r = (a >= PI_F) ? hi : r;
    return r;
}

// Eight per-lane accumulators summed over the wave in ~20 instructions instead of eight 6-step butterflies: every level
// halves the number of values a lane carries (v_permlane32_swap / v_permlane16_swap exchange half of one register
// against half of another, so one swap and one add serve two values), the last three levels are plain DPP adds.
// Returns, in lane l, the wave total of a[l >> 3].  The tree is fixed: results are deterministic.
__device__ __forceinline__ float reduce8_over_wave(float (&a)[8], int lane)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // lanes < 32 go on with values 0..3, lanes >= 32 with 4..7
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 4]), false, false);
        a[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // even rows of 16 lanes: value i, odd rows: value i + 2
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
        a[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    const bool hi8 = (lane & 8) != 0;
    const float keep = hi8 ? a[1] : a[0], send = hi8 ? a[0] : a[1];
    float v = keep + __uint_as_float(dpp_mov<0x128, 0xF>(__float_as_uint(send), __float_as_uint(send)));   // row_ror:8
    v += __uint_as_float(dpp_mov<0x141, 0xF>(__float_as_uint(v), __float_as_uint(v)));                     // row_half_mirror: l <-> 7 - l
    v += __uint_as_float(dpp_mov<0x1B, 0xF>(__float_as_uint(v), __float_as_uint(v)));                      // quad_perm [3,2,1,0]
    v += __uint_as_float(dpp_mov<0xB1, 0xF>(__float_as_uint(v), __float_as_uint(v)));                      // quad_perm [1,0,3,2]
    return v;
}

// the same for four accumulators: lane l ends with the wave total of a[l >> 4]
__device__ __forceinline__ float reduce4_over_wave(float (&a)[4], int lane)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
        a[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[0]), __float_as_uint(a[1]), false, false);
    float v = __uint_as_float(r.x) + __uint_as_float(r.y);
    v += __uint_as_float(dpp_mov<0x128, 0xF>(__float_as_uint(v), __float_as_uint(v)));                     // row_ror:8
    v += __uint_as_float(dpp_mov<0x141, 0xF>(__float_as_uint(v), __float_as_uint(v)));                     // row_half_mirror
    v += __uint_as_float(dpp_mov<0x1B, 0xF>(__float_as_uint(v), __float_as_uint(v)));                      // quad_perm [3,2,1,0]
    v += __uint_as_float(dpp_mov<0xB1, 0xF>(__float_as_uint(v), __float_as_uint(v)));                      // quad_perm [1,0,3,2]
    return v;
}

// how the (measurement chunk, feature group) grid of pass 1 is dealt to the 8 waves: chunks of 2^csh measurements (8, or
// 4 for small scans so that every wave has a chunk); NC = waves side by side over chunks (power of two <= 8), NF = 8 / NC
// waves over feature groups of 64
struct Pass1Grid { int csh, nchunks, NC, NF; };
__device__ __forceinline__ Pass1Grid pass1_grid(int M)
{
    Pass1Grid g;
    g.csh = (M <= 32) ? 2 : 3;
    g.nchunks = (M + (1 << g.csh) - 1) >> g.csh;
    g.NC = 1;
    while (g.NC < g.nchunks && g.NC < PHD_NW) g.NC <<= 1;
    g.NF = PHD_NW / g.NC;
    return g;
}

// Z_m's feature sum: the partial sums of the NF waves that own measurement m's chunk
__device__ __forceinline__ float pass1_feature_sum(const Lds& L, const Pass1Grid& g, int m, int M)
{
    // measurement m belongs to ONE chunk, chunk c to the NF waves (fg, c mod NC): NF partial sums per measurement, stored
    // [fg][M] — NF > 1 only for scans of at most 16 measurements, so NF M <= max(M, 32) floats (phd_lds.h)
    float s = L.zpart[m];
    for (int fg = 1; fg < g.NF; ++fg) s += L.zpart[fg * M + m];
    return s;
}

#define PHD_CAND_CAP ((4 * PHD_NW * 64 + 4 * 7 * 64) / 2)   // u16 entries of the candidate list (the part + win arrays)

// ---- pass 1 ------------------------------------------------------------------------------------------------
// For every measurement m: sum_j exp(lw_jm) over the in-range features (src/phdfilter.cu:2190-2223), and the list of
// the terms with lw >= c0m (the only ones that can survive the prune, see phd_kernels.hip).
//
// Lanes <-> features, a wave's chunk of 8 measurements in registers: the feature's terms (r, b, S, folded log-weight)
// are loaded ONCE, the measurement is wave-uniform, the eight sums are private accumulators, and the per-pair work is
// the arithmetic alone — no LDS traffic, no bounds tests (a lane past the last feature carries log-weight -inf), no
// cross-lane step.  The wave totals come from one transposing reduction per chunk.  Candidates: one bit per pair in a
// lane register, listed with one wave scan and one LDS atomic per 8 x 64 pairs.
template <bool FASTWRAP, int CH>
__device__ __forceinline__ u32 pass1_octet(const float (&zr)[CH], const float (&zb)[CH], u32 vm, const v4f fa, const v2f fc,
                                           float c0m, float (&acc)[CH])
{
    u32 bits = 0;
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        if ((vm >> q) & 1u) {                                     // uniform: measurement in range of M, label accepted
            const float i0 = zr[q] - fa.x;
            const float a = zb[q] - fa.y;
            const float i1 = FASTWRAP ? wrap_angle_small(a) : wrap_angle(a);
            const float dist = i0 * i0 * fa.z + i0 * i1 * fa.w + i1 * i1 * fc.x;                       // :1908-1910
            const float lw = fc.y - 0.5f * dist;
            acc[q] += __expf(lw);                                                                     // :2205
            bits |= !(lw < c0m) ? (1u << q) : 0u;                 // NaN stays a candidate, as in the dense test
        }
    }
    return bits;
}

// The common case — every measurement of the chunk valid, every |z_b| < pi — two measurements per packed instruction
// (v_pk_add / v_pk_mul / v_pk_fma: one issue slot, two pairs) and no branch between the pairs, so the compiler can
// interleave them.  wrap_angle's conditional "+- 2 pi" becomes r = fma(-A, k, a), r = fma(-B, k, r) with
// k = sign(a) [|a| >= pi]: a - A k is formed exactly and rounded once, like the subtraction it replaces (same bits).
template <int CH>
__device__ __forceinline__ u32 pass1_octet_packed(const float (&zr)[CH], const float (&zb)[CH], const v4f fa, const v2f fc,
                                                  float c0m, float (&acc)[CH])
{
    const float TWO_PI_F = 6.2831855f, B = -1.7484555e-7f;
    const v2f r2 = (v2f){fa.x, fa.x}, b2 = (v2f){fa.y, fa.y}, s00 = (v2f){fa.z, fa.z}, s01 = (v2f){fa.w, fa.w},
              s11 = (v2f){fc.x, fc.x};
    // the log-weight straight in base 2 (round 5: one packed multiply less per pair of measurements, +0.6 % at 4096 x 256 x 64):
    // l2 = lwb log2 e - (0.5 log2 e) d, candidates by l2 against c0m log2 e (c0m carries 1e-3 of slack for exactly such roundings)
    const float LOG2E = 1.44269504f;
    const float lwb2s = fc.y * LOG2E, c0m2 = c0m * LOG2E;
    const v2f lwb2 = (v2f){lwb2s, lwb2s};
    u32 bits = 0;
    const v2f c0m22 = (v2f){c0m2, c0m2};
#pragma unroll
    for (int q = CH - 2; q >= 0; q -= 2) {                  // descending: v_alignbit shifts the mask left, bit q ends at position q
        const v2f i0 = (v2f){zr[q], zr[q + 1]} - r2;
        const v2f a = (v2f){zb[q], zb[q + 1]} - b2;
        v2f k;
        // k = sign(a) [|a| >= pi] as trunc(a / pi) (round 5): with c = fl(1 / pi_f) = 0.31830987 the correctly rounded product fl(a c)
        // is monotone in a, 1.0 at a = pi_f (3.14159274) and 0.99999994 at the float below it, 1.9999999 at the largest difference
        // this path can see (|z_b|, |b| <= 3.1415925) - the same k for every float, three instructions per pair of measurements
        // instead of six (compare / copysign / select twice): +0.5 % at 4096 x 256 x 64, the same bits
        const v2f kt = a * (v2f){0.31830987f, 0.31830987f};
        k.x = __builtin_truncf(kt.x);
        k.y = __builtin_truncf(kt.y);
        v2f i1 = __builtin_elementwise_fma((v2f){-TWO_PI_F, -TWO_PI_F}, k, a);
        i1 = __builtin_elementwise_fma((v2f){-B, -B}, k, i1);
        v2f dist = (i0 * i0) * s00;                                                                    // :1908-1910
        dist = __builtin_elementwise_fma(i0 * i1, s01, dist);
        dist = __builtin_elementwise_fma(i1 * i1, s11, dist);
        const v2f l2 = __builtin_elementwise_fma(dist, (v2f){-0.5f * 1.44269504f, -0.5f * 1.44269504f}, lwb2);
        acc[q] += __builtin_amdgcn_exp2f(l2.x);                                                       // :2205 (= __expf up to the rounding of its argument)
        acc[q + 1] += __builtin_amdgcn_exp2f(l2.y);
        // candidate <=> !(l2 < c0m2) (a NaN stays a candidate, as in the dense test), without the compare -> scalar mask -> select -> or
        // chain and its wait states (7 of the block's 117 instructions were s_nop; round 5): min(l2, c0m2) is c0m2 exactly for a
        // candidate AND for a NaN (v_min returns the other operand), below it otherwise - so the SIGN BIT of min(l2, c0m2) - c0m2 is set
        // exactly for the terms that cannot survive, and two v_alignbit shift the pair's bits into the mask
        const v2f t = (v2f){fminf(l2.x, c0m2), fminf(l2.y, c0m2)} - c0m22;
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t.y), 31);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t.x), 31);
    }
    return ~bits & ((1u << CH) - 1u);
}

template <int CH>
__device__ __forceinline__ void pass1_chunks(const Lds& L, const Pass1Grid& g, int n_in, int M, int MM, int lane, int wave,
                                             bool sparse2, float c0m)
{
    const int cg = wave & (g.NC - 1), fg = wave / g.NC;
    lds_u16 clist = (lds_u16)L.part;
    for (int c = cg; c < g.nchunks; c += g.NC) {
        // chunk c = measurements c, c + nchunks, ...: strided, so that every chunk gets its share of the scan's dense part
        // (a scan lists what the sensor saw in sensor order: consecutive measurements tend to hit the same piece of map)
        const int mstep = g.nchunks;
        // the chunk's measurements (LDS broadcast reads: the same address in every lane)
        float zr[CH], zb[CH];
        u32 vmv = 0, slow = 0;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int m = c + q * mstep, mm = m < M ? m : M - 1;
            zr[q] = L.z_r[mm];
            zb[q] = L.z_b[mm];
            const bool ok = (m < M) && L.zok[mm];
            vmv |= ok ? (1u << q) : 0u;
            // the features' bearings lie in [-pi, pi] (wrap_angle's range): |z_b| < pi keeps every difference below 2 pi
            slow |= (ok && !(fabsf(zb[q]) < 3.14159274f)) ? 1u : 0u;
        }
        const u32 vm = (u32)__builtin_amdgcn_readfirstlane((int)vmv);
        const bool fast = __builtin_amdgcn_readfirstlane((int)slow) == 0;
        float acc[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) acc[q] = 0.f;
        // The candidate bits of up to 32 / CH feature groups wait in ONE register per lane and are listed together: the wave scan,
        // the atomic and the loop's set-up are paid once per chunk (at 256 in-range features: once instead of four times).
        constexpr int GPF = 32 / CH;                              // feature groups per flush
        u32 pend = 0;                                             // bit (slot * CH + q): measurement q of the chunk, group `slot` since the flush
        int slot = 0;
        const int jstep = g.NF * 64;
        auto flush = [&](int jb0) {                               // jb0: first group of the pending bits
            const int np = __popc(pend);
            const int incl = (int)wave_incl_scan((u32)np);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot) {                                            // uniform
                int base = 0;
                if (lane == 63) base = atomicAdd((int*)&L.ctr[CTR_NCAND], tot);
                int pos = __builtin_amdgcn_readlane(base, 63) + incl - np;
                while (pend) {
                    const int b = __builtin_ctz(pend);
                    pend &= pend - 1;
                    const int q = b & (CH - 1), sl = b / CH;
                    if (pos < PHD_CAND_CAP) clist[pos] = (u16)((c + q * mstep) * n_in + (jb0 + sl * jstep + lane));
                    ++pos;
                }
            }
            pend = 0;
        };
        int jb = fg * 64;
        for (; jb < n_in; jb += jstep) {
            const int j = jb + lane;
            const int jj = j < n_in ? j : n_in - 1;
            const v4f fa = L.f_a[jj];
            v2f fc = L.f_c[jj];
            if (j >= n_in) fc.y = -INFINITY;                      // contributes exp(-inf) = 0, never a candidate
#ifdef PHD_PASS1_RAGGED_R5
            const u32 bits = (fast && vm == (1u << CH) - 1u) ? pass1_octet_packed<CH>(zr, zb, fa, fc, c0m, acc)
                             : fast                          ? pass1_octet<true, CH>(zr, zb, vm, fa, fc, c0m, acc)
                                                             : pass1_octet<false, CH>(zr, zb, vm, fa, fc, c0m, acc);
#else
            // A RAGGED chunk — a scan whose length is not a multiple of the chunk (the last chunk of every real scan: 14 ... 44
            // measurements in the bundled data) or a measurement whose label is rejected — takes the packed form too (round 6): the
            // slots without a measurement carry a copy of the scan's last one (loaded above), their sums are dropped below and their
            // candidate bits masked here.  Until round 5 such a chunk ran the one-measurement-at-a-time form, about twice the
            // instructions, on ONE wave with the other seven waiting at the barrier behind it: +1.0 ... +1.5 % on scans of 27 / 44 / 61
            // measurements at 4096 x 256.  The packed form takes exp2 of a pre-scaled argument where the one-at-a-time form takes
            // __expf: the two agree to the rounding of that argument (an ulp of the sum, far inside the parity tolerances), so a
            // ragged chunk now rounds exactly like a full one — and a chained run (331 steps with resampling) is not bit-identical
            // to a round-5 build's.
            const u32 bits = fast ? (pass1_octet_packed<CH>(zr, zb, fa, fc, c0m, acc) & vm)
                                  : pass1_octet<false, CH>(zr, zb, vm, fa, fc, c0m, acc);
#endif
            if (sparse2) {
                pend |= bits << (slot * CH);
                if (++slot == GPF) { flush(jb - (GPF - 1) * jstep); slot = 0; }   // (uniform)
            }
        }
        if (sparse2 && slot) flush(jb - slot * jstep);
#ifndef PHD_PASS1_RAGGED_R5
        if (fast && vm != (1u << CH) - 1u) {                      // (uniform) the slots without a measurement: whatever their copies summed up is dropped
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (!((vm >> q) & 1u)) acc[q] = 0.f;
        }
#endif
        float tot;
        int m;
        if (CH == 8) { tot = reduce8_over_wave((float (&)[8])acc, lane); m = c + (lane >> 3) * mstep; }
        else { tot = reduce4_over_wave((float (&)[4])acc, lane); m = c + (lane >> 4) * mstep; }
        if ((lane & (64 / CH - 1)) == 0 && m < M) L.zpart[fg * M + m] = tot;
    }
}

__device__ __forceinline__ void pass1_normalisers(const Lds& L, int n_in, int M, int MM, int tid, bool sparse2, float c0m)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Pass1Grid g = pass1_grid(M);
    if (g.csh == 3) pass1_chunks<8>(L, g, n_in, M, MM, lane, wave, sparse2, c0m);
    else pass1_chunks<4>(L, g, n_in, M, MM, lane, wave, sparse2, c0m);
}

} // namespace phd
