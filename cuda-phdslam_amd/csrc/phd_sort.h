// phd_sort.h — survivor sorts of the merge: register bitonic network, rank by counting, counting sort on the weight key.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"

namespace phd {

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
// the far-pair filter's per-Gaussian term (see Lds::gA): pair (a, b) can be within the merge distance only if
//   |ma - mb|^2 < 0.505 T (tr Pa + tr Pb)   <=>   E_a + E_b - 2 ma.mb < 0,   E = |m|^2 - 0.505 T tr P
// (d >= 2 |dm|^2 / (tr Pa + tr Pb) for SPD covariances; 1 % guard band).  The expanded form costs one add and two FMAs
// per pair; its cancellation error is below 8 eps (|ma|^2 + |mb|^2), covered by shrinking |m|^2 by 2e-6.  Not SPD:
// tr = +inf -> E = -inf -> always a candidate.  Tpre = 1.01 T, or -1 when T <= 0 (never a candidate, like the exact test).
__device__ __forceinline__ float filter_term(float mx, float my, float xx, float xy, float yy, float Tpre, bool hellinger)
{
    const bool spd = (xx > 0.f) && (yy > 0.f) && (xx * yy - xy * xy > 0.f);
    const float tr = spd ? (xx + yy) : INFINITY;
    if (hellinger) return -INFINITY;                    // the Hellinger metric has no cheap filter: every pair is a candidate
    return (mx * mx + my * my) * (1.f - 2e-6f) - 0.5f * Tpre * tr;
}

// sorted SoA planes (what the fall-back sorts leave) -> the float4 arrays gA / gB, in place
__device__ __forceinline__ void planes_to_aos(const Lds& L, int S, int tid, float Tpre, bool hellinger)
{
    float rw[4], rmx[4], rmy[4], rxx[4], rxy[4], ryy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        rw[e] = rmx[e] = rmy[e] = rxx[e] = rxy[e] = ryy[e] = 0.f;
        if (i < S) { rw[e] = L.w[i]; rmx[e] = L.mx[i]; rmy[e] = L.my[i]; rxx[e] = L.xx[i]; rxy[e] = L.xy[i]; ryy[e] = L.yy[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        if (i < S) {
            L.gA[i] = (v4f){rmx[e], rmy[e], filter_term(rmx[e], rmy[e], rxx[e], rxy[e], ryy[e], Tpre, hellinger), rw[e]};
            L.gB[i] = (v4f){rxx[e], rxy[e], ryy[e], __int_as_float(-1)};
        }
    }
    __syncthreads();
}

// (writes the sorted survivors as the float4 arrays gA / gB — see Lds)
// The bucket counts arrive finished in the `tr` plane: every emission site counted its survivor (store_survivor, phd_lds.h).
__device__ __forceinline__ bool bucket_sort_survivors(const Lds& L, int S, int S_cap, int tid, int lane, int wave, int n_update,
                                                      float Tpre, bool hellinger, const BucketMap& bm)
{
    lds_u32 cntc = (lds_u32)L.tr;     // per bucket: count -> (placed << 16) | start
    const int NB = S_cap;             // buckets: a power of two >= 512
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
        if (bmax > 128u) { __syncthreads(); return false; }       // uniform: crowded bucket (many equal weights) -> the network
        u32 run = woff + incl - local;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < per) { cntc[lo + e] = run; run += v[e]; }
    }
    __syncthreads();
    // keys (weight | ~slab index), stored IN BUCKET ORDER (arrival order inside a bucket): the rank loop then reads its
    // bucket's keys straight from consecutive slots — one LDS round trip per step instead of index -> key
    LDS_T(u64)* const bkeys = (LDS_T(u64)*)L.khi;              // [S]: the khi and klo planes are adjacent
    u64 mine[4];
    int bk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        mine[e] = 0ull; bk[e] = 0;
        if (i < S) {
            const int u0 = L.u[i];
            const float w = L.w[i];
            mine[e] = ((u64)orderable(w) << 32) | (0xFFFFFFFFu - (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0));
            bk[e] = bucket_of(bm, w);
            const u32 old = atomicAdd((u32*)&cntc[bk[e]], 0x10000u);
            bkeys[(old & 0xFFFFu) + (old >> 16)] = mine[e];
        }
    }
    __syncthreads();
    // rank inside the bucket = keys above mine (keys are unique: (weight, slab index)).  A thread's entries e, e + 1 walk
    // their buckets together, two keys each per trip: four independent reads in flight (the walk is LDS latency)
    int rank[4];
#pragma unroll
    for (int ep = 0; ep < 4; ep += 2) {
        rank[ep] = rank[ep + 1] = 0;
        if (ep * PHD_T >= S) continue;                            // uniform
        const bool v0 = tid + ep * PHD_T < S, v1 = tid + (ep + 1) * PHD_T < S;
        const u32 c0 = v0 ? cntc[bk[ep]] : 0u, c1 = v1 ? cntc[bk[ep + 1]] : 0u;
        const int b0 = (int)(c0 & 0xFFFFu), k0 = (int)(c0 >> 16), b1 = (int)(c1 & 0xFFFFu), k1 = (int)(c1 >> 16);
        const int kk = k0 > k1 ? k0 : k1;
        int r0 = 0, r1 = 0;
        for (int t = 0; t < kk; t += 2) {
            // (reads past a bucket's end fall into the next buckets or the slack behind the list: masked, never used)
            const u64 a0 = bkeys[b0 + t], a1 = bkeys[b0 + t + 1], d0 = bkeys[b1 + t], d1 = bkeys[b1 + t + 1];
            r0 += ((t < k0 && a0 > mine[ep]) ? 1 : 0) + ((t + 1 < k0 && a1 > mine[ep]) ? 1 : 0);
            r1 += ((t < k1 && d0 > mine[ep + 1]) ? 1 : 0) + ((t + 1 < k1 && d1 > mine[ep + 1]) ? 1 : 0);
        }
        rank[ep] = b0 + r0;
        rank[ep + 1] = b1 + r1;
    }
    // the planes are staged through registers only now (short live ranges: the kernel sits at its register budget)
    float rw[4], rmx[4], rmy[4], rxx[4], rxy[4], ryy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        rw[e] = rmx[e] = rmy[e] = rxx[e] = rxy[e] = ryy[e] = 0.f;
        if (i < S) { rw[e] = L.w[i]; rmx[e] = L.mx[i]; rmy[e] = L.my[i]; rxx[e] = L.xx[i]; rxy[e] = L.xy[i]; ryy[e] = L.yy[i]; }
    }
    __syncthreads(); // every read of the old order (and of the counts, which sit in the tr plane) precedes the writes below
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + e * PHD_T;
        if (i < S) {
            const int k = rank[e];
            L.gA[k] = (v4f){rmx[e], rmy[e], filter_term(rmx[e], rmy[e], rxx[e], rxy[e], ryy[e], Tpre, hellinger), rw[e]};
            L.gB[k] = (v4f){rxx[e], rxy[e], ryy[e], __int_as_float(-1)}; // unassigned
        }
    }
    __syncthreads();
    return true;
}

} // namespace phd
