// phd_lds.h — LDS layout of the update+merge kernel, its counters, survivor slot allocation.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"

namespace phd {

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

// merge rounds: window copy (pos[64] | gA[64] | gB[64]) + the round's seed records (2 x 72 float4), behind the sort arrays
#define PHD_RWIN_BYTES (4u * 64u + 2u * 16u * 64u + 2u * 16u * 72u)

struct LdsOffsets {
    u32 w, mx, my, xx, xy, yy, tr, u;
    u32 alias;      // start of the aliased region
    u32 cinfo;      // (inside it) the merge's per-cluster seed records, behind both the rounds' working set and the sums
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
    const u32 sort1 = 3u * sv + PHD_RWIN_BYTES;   // sort arrays / round lists + the rounds' window and seed records
    // the exact moment sums of the round-based merge (phd_fixsum.h): 48 B of accumulators per output cluster (<= C) from the
    // start of the region once the rounds are over, and 16 B per cluster of seed records written DURING the rounds — behind both
    const u32 sums = 48u * (u32)C;
    o.cinfo = o.alias + (sort1 > sums ? sort1 : sums);
    u32 amax = feat > sort1 ? feat : sort1;
    const u32 msum = (o.cinfo - o.alias) + 16u * (u32)C;
    amax = amax > msum ? amax : msum;
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
    // the same 32 S bytes as two float4 arrays, the layout of the SORTED survivors in the round-based merge (S > 256):
    //   gA[i] = (mean x, mean y, E, weight)   E = |mean|^2 (1 - 2e-6) - 0.505 T tr: the far-pair filter's per-Gaussian term
    //   gB[i] = (cov xx, cov xy, cov yy, cluster assignment as int bits)
    // one ds_read_b128 per use instead of three to six ds_read_b32 with their address arithmetic
    LDS_T(v4f)* gA;
    LDS_T(v4f)* gB;
    // aliased region
    LDS_T(v4f)* f_a;                           // per in-range feature: (r, b, S00, S01+S10)
    LDS_T(v2f)* f_c;                           //                       (S11, folded log-weight base)
    lds_u16 f_idx;                                // map index of in-range feature j
    LDS_T(v4f)* f_k;                              // Kalman gain K0..K3 of in-range feature j
    LDS_T(v4f)* f_p;                              // Joseph-form updated covariance (the same for every measurement) and prior mean x: (xx, xy, yy, mx)
    lds_f32 f_my;                                 // prior mean y
    lds_u32 khi, klo, pay;                        // sort 1
    LDS_T(long long)* acc;                        // exact moment sums: [C][6] 64-bit words (after the rounds; aliases khi.. )
    LDS_T(v4f)* cinfo;                            // per cluster: (seed mean x, y, seed weight, seed's survivor index), written in the rounds
    lds_f32 rwin;                                 // merge rounds: window copy + seed records (PHD_RWIN_BYTES), behind khi/klo/pay
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
    lds_f32 win;                      // 7*64 floats: with `part` the candidate list of pass 1 / 2
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
    L.gA = (LDS_T(v4f)*)(base + o.w); L.gB = (LDS_T(v4f)*)(base + o.xy);
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
    L.acc = (LDS_T(long long)*)(base + o.alias);
    L.cinfo = (LDS_T(v4f)*)(base + o.cinfo);
    L.rwin = (lds_f32)(base + o.alias + 3u * sv);
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


enum { CTR_NSURV = 0, CTR_NIN = 1, CTR_NOUT = 2, CTR_OVERFLOW = 3, CTR_KOUT = 4, CTR_NHEAD = 5, CTR_TMP = 6 /* ..+PHD_NW*2 <= 22 */,
       CTR_NPAIR = 24 /* merge_small: candidate pairs listed for the exact closeness test */,
       CTR_NCAND = 25 /* pass 1: terms listed for pass 2 */, CTR_WSYNC = 26 /* CPHD block: arrival counter of waves_sync */,
       CTR_NNEAR = 30 };

// The merge's first sort is a counting sort on the weight key (phd_sort.h).  Its bucket of a survivor is a FIXED function of
// the weight — the orderable bit pattern of a float is a log scale: (key(64) - key(w)) >> shift, clamped, with the shift
// chosen from min_feature_weight so that the weights the prune lets through spread over the S buckets — so every emission
// site counts its survivor's bucket as it stores it (one LDS atomic), and the sort starts from finished counts: no key pass,
// no range reduction, no counting pass (two barriers less).  The counts live in the `tr` plane, unused until the sort.
struct BucketMap { u32 kref; int shift; int nb; };
__device__ __forceinline__ u32 orderable_key(float w)
{
    const u32 b = __float_as_uint(w);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ BucketMap bucket_map(float min_weight, int S)
{
    BucketMap m;
    m.nb = S;
    m.kref = orderable_key(64.f);
    const float wlo = (min_weight > 1e-30f) ? min_weight * 0.125f : 1e-12f;
    const u32 range = m.kref - orderable_key(wlo);
    const int bits = 32 - __clz((int)range), lognb = 31 - __clz(S);
    m.shift = bits > lognb ? bits - lognb : 0;
    return m;
}
__device__ __forceinline__ int bucket_of(const BucketMap& m, float w)
{
    const u32 k = orderable_key(w);
    const u32 d = k > m.kref ? 0u : (m.kref - k) >> m.shift;
    return d < (u32)m.nb ? (int)d : m.nb - 1;
}

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

// Survivors past the LDS capacity (dense scans of a large map: up to G (M + 1) + M update components, a few per cent of
// which survive) go to a per-particle record list in HBM — records of 8 floats (w, mx, my, xx, xy, yy, slab index, -), list
// position = slot — and the particle's merge is done by phd_merge_spill_kernel instead of the LDS merge (phd_spill.h).
struct SpillRef {
    float* rec;     // this particle's records (NULL: no spill path — overflow is an error)
    int cap;        // records the list holds (LDS survivors included: they are copied in before the hand-over)
};

__device__ __forceinline__ void spill_store(const SpillRef& sp, const Lds& L, int slot, float w, float mx, float my, float xx,
                                            float xy, float yy, int u)
{
    if (sp.rec && slot < sp.cap) {
        float* r = sp.rec + (size_t)slot * 8;
        r[0] = w; r[1] = mx; r[2] = my; r[3] = xx; r[4] = xy; r[5] = yy; r[6] = __int_as_float(u); r[7] = 0.f;
    } else {
        L.ctr[CTR_OVERFLOW] = 1;
    }
}

__device__ __forceinline__ void store_survivor(const Lds& L, int slot, int S, float w, float mx, float my, float xx,
                                               float xy, float yy, int u, const SpillRef& sp, const BucketMap& bm)
{
    if (slot < S) {
        L.w[slot] = w; L.mx[slot] = mx; L.my[slot] = my;
        L.xx[slot] = xx; L.xy[slot] = xy; L.yy[slot] = yy;
        L.u[slot] = u;
        __hip_atomic_fetch_add((LDS_T(u32)*)L.tr + bucket_of(bm, w), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        spill_store(sp, L, slot, w, mx, my, xx, xy, yy, u);
    }
}

} // namespace phd
