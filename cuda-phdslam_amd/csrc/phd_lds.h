// phd_lds.h — LDS layout of the update+merge kernel, its counters, survivor slot allocation.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds_layout.h"

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

struct Lds {
    // survivors (SoA), S entries each
    lds_f32 w, mx, my, xx, xy, yy;
    lds_f32 tr; // trace of the covariance (+inf if not SPD): cheap far-pair filter of the merge
    lds_i32 u; // slab index (sort tie-break); after the sort: cluster assignment
    // the same 32 S bytes as two float4 arrays, the layout of the SORTED survivors in the round-based merge (S > 256):
    //   gA[i] = (mean x, mean y, E, weight)   E = |mean|^2 (1 - 2e-6) - 0.505 T tr: the far-pair filter's per-Gaussian term
    //   gB[i] = (cov xx, cov xy, cov yy, assignment word)   assignment = cluster index | the cluster's seed's index << 16, -1: none
    // one ds_read_b128 per use instead of three to six ds_read_b32 with their address arithmetic
    LDS_T(v4f)* gA;
    LDS_T(v4f)* gB;
    // region X
    LDS_T(v4f)* f_a;                           // per in-range feature: (r, b, S00, S01+S10)
    LDS_T(v2f)* f_c;                           //                       (S11, folded log-weight base)
    lds_u16 f_idx;                                // map index of in-range feature j
    lds_u32 khi, klo, pay;                        // sort 1
    lds_u16 ulist;                                // merge rounds: two u16 lists of S entries (the unmerged survivors)
    lds_f32 rwin;                                 // merge rounds: window copy + seed records (PHD_RWIN_BYTES), behind the lists
    LDS_T(long long)* acc;                        // exact moment sums: 48 B per cluster from the start of X to wide_end
    int acc_clusters;                             // clusters the accumulators hold at a time
    LDS_T(u64)* srow;                             // merge_small: [256][4] closeness to earlier positions
    LDS_T(u64)* sseed;                            // merge_small: [4] seed mask
    LDS_T(v4f)* sA;                               // merge_small: [256] (mx, my, 0.505 T tr, w) in sorted order
    LDS_T(v4f)* sB;                               // merge_small: [256] (xx, xy, yy, assignment word)
    lds_u32 splist;                               // merge_small: candidate pairs, in the dead survivor planes
    int splist_cap;
    // behind X (dead once the update / the rounds are over)
    lds_u32 part;                     // 4*64 row parts of the window closeness matrix
    lds_f32 win;                      // 7*64 floats: with `part` the candidate list of pass 1 / 2
    lds_f32 z_r, z_b, logZ;           // MM each
    lds_u32 zok;                      // MM
    lds_f32 zpart;                    // max(MM, 32): pass 1's partial sums (CPHD: redirected to the block's own scratch)
    // persistent
    lds_u16 out_idx;                  // C
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
    L.f_idx = (lds_u16)(base + f);
    const u32 sv = align16u(4u * (u32)S);
    L.khi = (lds_u32)(base + o.alias);
    L.klo = (lds_u32)(base + o.alias + sv);
    L.pay = (lds_u32)(base + o.alias + 2u * sv);
    L.ulist = (lds_u16)(base + o.alias);
    L.rwin = (lds_f32)(base + o.alias + sv);
    L.acc = (LDS_T(long long)*)(base + o.alias);
    L.acc_clusters = (int)((o.wide_end - o.alias) / 48u);
    L.srow = (LDS_T(u64)*)(base + o.alias);
    L.sseed = (LDS_T(u64)*)(base + o.alias + PHD_SMALL_S * 32u);
    if (S >= 4 * PHD_SMALL_S) {
        // the planes are dead once the sorted staging has read them: records in the first two, pairs in the next three
        L.sA = (LDS_T(v4f)*)(base + o.w);
        L.sB = (LDS_T(v4f)*)(base + o.mx);
        L.splist = (lds_u32)(base + o.my);
        L.splist_cap = 3 * S;
    } else {
        L.sA = (LDS_T(v4f)*)(base + o.alias + 48u * PHD_SMALL_S);
        L.sB = (LDS_T(v4f)*)(base + o.alias + 48u * PHD_SMALL_S + PHD_SMALL_S * 16u);
        L.splist = (lds_u32)(base + o.w);
        L.splist_cap = 5 * S;
    }
    L.part = (lds_u32)(base + o.part);
    L.win = (lds_f32)(base + o.win);
    L.z_r = (lds_f32)(base + o.z_r); L.z_b = (lds_f32)(base + o.z_b); L.logZ = (lds_f32)(base + o.logZ);
    L.zok = (lds_u32)(base + o.zok);
    L.zpart = (lds_f32)(base + o.zpart);
    L.out_idx = (lds_u16)(base + o.out_idx);
    L.red = (lds_f32)(base + o.red);
    L.ctr = (lds_i32)(base + o.ctr);
    return L;
}


enum { CTR_NSURV = 0, CTR_NIN = 1, CTR_NOUT = 2, CTR_OVERFLOW = 3, CTR_KOUT = 4, CTR_NHEAD = 5, CTR_TMP = 6 /* ..+PHD_NW*2 <= 22 */,
       CTR_NPAIR = 24 /* merge_small: candidate pairs listed for the exact closeness test */,
       CTR_NCAND = 25 /* pass 1: terms listed for pass 2 */, CTR_WSYNC = 26 /* CPHD block: arrival counter of waves_sync */, CTR_WSYNC2 = 27 /* ... of the cardinality update that runs beside the chains */,
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
