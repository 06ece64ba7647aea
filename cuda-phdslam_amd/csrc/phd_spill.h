// phd_spill.h — the merge of a particle whose survivor list does not fit LDS (more than S_cap pruned update components:
// dense scans of large maps — the reference allows M = 256 measurements and has no cap on the map, src/phdfilter.cu:3390-3394,
// src/main.cpp:1003).  Part of the one translation unit phd_kernels.hip (device code, namespace phd).
//
// The update kernel writes such a particle's survivors to a record list in HBM (SpillRef, phd_lds.h) and leaves the merge
// to this kernel: the same greedy algorithm (seed = heaviest unmerged survivor, ties to the lowest slab index; absorb
// d < T; moment matching in weight order; stop at W == 0 — src/phdfilter.cu:2739-2890, src/gm_reduce.cpp:57-134) in its
// round-based form on global memory: rank by counting (keys tiled through LDS), then rounds of up to 64 seeds with the
// window and the round's seed records in LDS, the survivors and the unmerged lists in HBM.  Correct for any list length up
// to spill_cap; LDS-resident lists take merge_in_lds.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_merge.h"

namespace phd {

#define PHD_SPILL_TILE 2048

// 64-bit accumulators of the exact moment sums in global memory (L2 atomics; reads bypass the L1)
__device__ __forceinline__ void g_add64(long long* p, long long v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long g_ld64(const long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned g_ld32(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Round 3: the same ROUNDS as merge_in_lds (phd_merge.h) — a window of the 64 heaviest unmerged survivors, their pairwise
// closeness, seeds by the fixed point of s_k = not exists l < k: close(k, l) and s_l, every other unmerged survivor tested
// against the round's seeds in order (sign-bit far-pair filter, exact decision on the marked ones) and joining the first it
// is close to, ordered compaction of the list — with the survivors and the lists in HBM (L2-resident: a particle's records
// are ~100 KB) and only the window and the round's seed records in LDS; then the exact, order-free moment sums (phd_fixsum.h)
// on 64-bit accumulators in HBM.  Round 2 took one seed per trip (three barriers and three dependent global round trips for
// each of the ~350 clusters of a dense scan, then one thread per cluster scanning all assignments): 3 ms per particle.
template <bool HELLINGER>
__device__ __forceinline__ void merge_spill_body(const UpdateArgs& A, int p, int S, int n_update, int n_out0, float r_out_scale, int src)
{
    __shared__ u64 s_keys[PHD_SPILL_TILE];
    __shared__ v4f s_wA[64], s_wB[64];          // window: (mean x, mean y, E, weight), (cov xx, xy, yy, -)
    __shared__ v4f s_sF[64], s_sG[64];          // the round's seeds in order: (-2 mx, -2 my, E, cluster index), (cov, -)
    __shared__ u64 s_row[64];
    __shared__ int s_wpos[64];
    __shared__ int s_scan[PHD_NW];
    __shared__ int s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DevConfig& cfg = A.cfg;
    const float T = cfg.minSeparation;
    const float Tpre = (T > 0.f) ? T * 1.01f : -1.f;
    const int cap = A.cap;
    float* rec = A.spill_rec + (size_t)p * 2 * A.spill_cap * 8;        // arrival order
    float* srt = rec + (size_t)A.spill_cap * 8;                        // (weight desc, slab index asc) order
    int* list_a = (int*)rec;                                           // reused once the sorted copy exists: the unmerged lists
    int* list_b = list_a + A.spill_cap;
    long long* acc = A.spill_acc + (size_t)p * cap * 8;                // per cluster: 6 words of sums + the seed's record (16 B)
    const unsigned rows_stride = A.fuse_weights ? 0u : A.out_stride;
    float* out = A.map_out + (size_t)p * (rows_stride ? rows_stride : (size_t)6 * cap);
    // src: the slab the update kernel read (handed over in spill_meta: parent[p] has been reset to p by then)
    const float* in = A.map_in + (size_t)src * 6 * cap;

    // ---- sort by (weight desc, slab index asc): a counting sort on the leading bits of the weight key (the orderable bit
    //      pattern of a float is a log scale, on which the weights are nearly uniform: buckets of a few survivors), rank =
    //      bucket start + larger keys in the own bucket.  (Round 2 ranked by counting over ALL keys: S^2 / 512 comparisons per
    //      thread, 0.4 ms at 2 700 survivors.)  Equal weights crowd a bucket and cost O(bucket^2): results unaffected. ----
    auto key_of = [&](int i) -> u64 {
        const int u0 = __float_as_int(rec[(size_t)i * 8 + 6]);
        const u32 ml = 0xFFFFFFFFu - (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0);
        return ((u64)orderable(rec[(size_t)i * 8 + 0]) << 32) | ml;
    };
    u32* const cnt = (u32*)s_keys;                                     // [2 * PHD_SPILL_TILE] buckets: count -> (placed << 16) | start
    const int NB = 2 * PHD_SPILL_TILE;
    int* const members = A.spill_tmp + (size_t)p * A.spill_cap;       // bucket segments in arrival order
    u32 kmn = 0xFFFFFFFFu, kmx = 0u;
    for (int i = tid; i < S; i += PHD_T) {
        const u32 k = orderable(rec[(size_t)i * 8 + 0]);
        kmn = k < kmn ? k : kmn;
        kmx = k > kmx ? k : kmx;
    }
    for (int b = tid; b < NB; b += PHD_T) cnt[b] = 0u;
    kmn = (u32)~wave_max_i((int)~kmn ^ (int)0x80000000) ^ 0x80000000u;  // (unsigned min/max through the signed wave maximum)
    kmx = (u32)wave_max_i((int)(kmx ^ 0x80000000u)) ^ 0x80000000u;
    if (lane == 0) { s_scan[wave] = (int)kmn; s_row[wave] = (u64)kmx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < PHD_NW; ++w) {
        const u32 a0 = (u32)s_scan[w], b0 = (u32)s_row[w];
        kmn = a0 < kmn ? a0 : kmn;
        kmx = b0 > kmx ? b0 : kmx;
    }
    const u32 range = kmx - kmn;
    const int bits = range ? 32 - __clz((int)range) : 0;
    const int shift = bits > 12 ? bits - 12 : 0;                       // (range >> shift) < 4096 = NB; bucket 0 holds the largest weights
    __syncthreads();
    for (int i = tid; i < S; i += PHD_T) atomicAdd(&cnt[(kmx - orderable(rec[(size_t)i * 8 + 0])) >> shift], 1u);
    __syncthreads();
    {
        const int lo = tid * (NB / PHD_T);                             // 8 buckets per thread
        u32 v[NB / PHD_T], local = 0u;
#pragma unroll
        for (int e = 0; e < NB / PHD_T; ++e) { v[e] = cnt[lo + e]; local += v[e]; }
        const u32 incl = wave_incl_scan(local);
        if (lane == 63) s_scan[wave] = (int)incl;
        __syncthreads();
        u32 run = incl - local;
#pragma unroll
        for (int w = 0; w < PHD_NW; ++w)
            if (w < wave) run += (u32)s_scan[w];
#pragma unroll
        for (int e = 0; e < NB / PHD_T; ++e) { cnt[lo + e] = run; run += v[e]; }
    }
    __syncthreads();
    for (int i = tid; i < S; i += PHD_T) {
        const u32 old = atomicAdd(&cnt[(kmx - orderable(rec[(size_t)i * 8 + 0])) >> shift], 0x10000u);
        members[(old & 0xFFFFu) + (old >> 16)] = i;
    }
    __syncthreads();
    __threadfence_block();
    for (int i = tid; i < S; i += PHD_T) {
        const float* r = rec + (size_t)i * 8;
        const u32 c = cnt[(kmx - orderable(r[0])) >> shift];
        const int base = (int)(c & 0xFFFFu), k = (int)(c >> 16);
        const u64 mine = key_of(i);
        int rank = base;
        for (int t = 0; t < k; ++t) rank += (key_of(members[base + t]) > mine) ? 1 : 0;   // keys are unique: (weight, slab index)
        float* d = srt + (size_t)rank * 8;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) d[k2] = r[k2];
        d[6] = __int_as_float(-1);                                      // cluster index: unassigned
        d[7] = filter_term(r[1], r[2], r[3], r[4], r[5], Tpre, HELLINGER);
    }
    __syncthreads();                       // (block-scope visibility of the global writes: same CU, L1 write-through + barrier)
    __threadfence_block();
    for (int i = tid; i < S; i += PHD_T) list_a[i] = i;
    if (tid == 0) s_stop = 0x7FFFFFFF;
    __syncthreads();

    // ---- rounds ----
    int* cur = list_a;
    int* nxt = list_b;
    int n_u = S, kbase = 0;
    while (n_u > 0) {
        const int nwin = n_u < 64 ? n_u : 64;
        const int nrest = n_u - nwin;
        if (tid < 64) {
            const int i = cur[tid < nwin ? tid : nwin - 1];
            const float* r = srt + (size_t)i * 8;
            s_wpos[tid] = i;
            s_wA[tid] = (v4f){r[1], r[2], tid < nwin ? r[7] : INFINITY, r[0]};
            s_wB[tid] = (v4f){r[3], r[4], r[5], 0.f};
            s_row[tid] = 0ull;
        }
        __syncthreads();
        // (1) window closeness: pair (k, l < k), eight pairs per thread
        for (int q = tid; q < 64 * 64; q += PHD_T) {
            const int k = q >> 6, l = q & 63;
            if (l < k && k < nwin) {
                const v4f ka = s_wA[k], la = s_wA[l];
                const float f = (ka.z + la.z) - 2.f * (ka.x * la.x + ka.y * la.y);
                if (f < 0.f || !(f == f)) {                             // (NaN: decided by the exact test)
                    const v4f kb = s_wB[k], lb = s_wB[l];
                    if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, ka.x, ka.y, kb.x, kb.y, kb.z, T))
                        atomicOr((unsigned long long*)&s_row[k], 1ull << l);
                }
            }
        }
        __syncthreads();
        // (2) seeds: every wave computes the same mask
        const u64 row = s_row[lane];
        const u64 live = (nwin == 64) ? ~0ull : ((1ull << nwin) - 1ull);
        u64 seeds = live;
        for (int it = 0; it < 65; ++it) {
            const u64 blocked = __ballot((row & seeds) != 0ull);
            const u64 nx = live & ~blocked;
            if (nx == seeds) break;
            seeds = nx;
        }
        const int nseeds = __popcll(seeds);
        if (wave == 0) {
            if (lane < nwin) {
                const bool is_seed = (seeds >> lane) & 1ull;
                const int owner = is_seed ? lane : __builtin_ctzll(row & seeds);
                const int c = kbase + __popcll(seeds & ((1ull << owner) - 1ull));
                srt[(size_t)s_wpos[lane] * 8 + 6] = __int_as_float(c);
                if (is_seed) {
                    const int rank = __popcll(seeds & lanemask_lt());
                    const v4f ka = s_wA[lane];
                    s_sF[rank] = (v4f){-2.f * ka.x, -2.f * ka.y, ka.z, __int_as_float(c)};
                    s_sG[rank] = s_wB[lane];
                    if (c < cap) {                                      // the cluster's record for the moment sums
                        float* ci = (float*)(acc + (size_t)c * 8 + 6);
                        ci[0] = ka.x; ci[1] = ka.y; ci[2] = ka.w; ci[3] = __int_as_float(s_wpos[lane]);
                    }
                }
            }
        }
        __syncthreads();
        // (3) the rest of the list: thread t owns the contiguous entries [t per, (t + 1) per)
        const int per = (nrest + PHD_T - 1) / PHD_T;                   // <= 64 (spill_cap <= 32768)
        int kept = 0;
        u64 keepbits = 0ull;
        for (int q = 0; q < per; ++q) {
            const int e = tid * per + q;
            if (e >= nrest) break;
            const int i = cur[64 + e];
            const float* r = srt + (size_t)i * 8;
            const float emx = r[1], emy = r[2], exx = r[3], exy = r[4], eyy = r[5], eE = r[7];
            bool merged = false;
            for (int sd = 0; sd < nseeds; ++sd) {
                const v4f f = s_sF[sd];
                const float t = (f.z + eE) + (f.x * emx + f.y * emy);
                if (t < 0.f || !(t == t)) {
                    const v4f g = s_sG[sd];
                    if (is_close<HELLINGER>(-0.5f * f.x, -0.5f * f.y, g.x, g.y, g.z, emx, emy, exx, exy, eyy, T)) {
                        srt[(size_t)i * 8 + 6] = f.w;
                        merged = true;
                        break;
                    }
                }
            }
            if (!merged) { kept++; keepbits |= 1ull << q; }
        }
        // ordered compaction
        const int incl = (int)wave_incl_scan((u32)kept);
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PHD_NW; ++w) {
            const int c = s_scan[w];
            if (w < wave) woff += c;
            total += c;
        }
        int o = woff + incl - kept;
        for (int q = 0; q < per; ++q)
            if ((keepbits >> q) & 1ull) nxt[o++] = cur[64 + tid * per + q];
        kbase += nseeds;
        n_u = total;
        __syncthreads();
        __threadfence_block();
        int* t2 = cur; cur = nxt; nxt = t2;
    }
    const int n_clusters = kbase;
    const int K = n_clusters < cap ? n_clusters : cap;

    // ---- moment matching: exact, order-free sums (phd_fixsum.h) on accumulators in HBM ----
    for (int t = tid; t < 6 * K; t += PHD_T) acc[(size_t)(t / 6) * 8 + (t % 6)] = 0;
    __syncthreads();
    __threadfence_block();
    for (int i = tid; i < S; i += PHD_T) {
        const float* r = srt + (size_t)i * 8;
        const int c = __float_as_int(r[6]);
        if (c < 0 || c >= K) continue;
        const float* ci = (const float*)(acc + (size_t)c * 8 + 6);
        const int Fw = fx_field(ci[2]);
        unsigned* fl = (unsigned*)(acc + (size_t)c * 8 + 5);            // word 10: largest exponent, word 11: flags
        if (__float_as_int(ci[3]) == i) {
            if (!seed_close_to_itself<HELLINGER>(r[1], r[2], r[3], r[4], r[5], T)) { atomicOr(fl + 1, 2u); continue; }
        }
        FxSums fs = {0, 0, 0, 0, 0, 0, 0, 0, 0, true};
        fx_add_first(fs, Fw, ci[0], ci[1], r[0], r[1], r[2], r[3], r[4], r[5]);
        long long* q = acc + (size_t)c * 8;
        g_add64(q + 0, fs.W); g_add64(q + 1, fs.xh); g_add64(q + 2, fs.xl); g_add64(q + 3, fs.yh); g_add64(q + 4, fs.yl);
        atomicMax(fl, (unsigned)fs.ec);
        if (!fs.ok) atomicOr(fl + 1, 1u);
    }
    __syncthreads();
    for (int c = tid; c < K; c += PHD_T) {
        long long* q = acc + (size_t)c * 8;
        FxSums fs;
        fs.W = g_ld64(q + 0); fs.xh = g_ld64(q + 1); fs.xl = g_ld64(q + 2); fs.yh = g_ld64(q + 3); fs.yl = g_ld64(q + 4);
        unsigned* fl = (unsigned*)(q + 5);
        const unsigned ec = g_ld32(fl), flg = g_ld32(fl + 1);
        const bool ok = !(flg & 1u) && ec < 255u, selfok = !(flg & 2u);
        const int Fw = fx_field(((const float*)(q + 6))[2]);
        int stop_at = 0x7FFFFFFF;
        if (fs.W == 0 && ok) stop_at = c;                               // src/phdfilter.cu:2821
        else if (!selfok) stop_at = c + 1;
        if (stop_at != 0x7FFFFFFF) atomicMin(&s_stop, stop_at);
        float W, mx, my;
        fx_mean(fs, Fw, W, mx, my);
        const int Fc = fx_cov_anchor(Fw, (int)ec);
        __hip_atomic_store(q + 0, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 2, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 3, fs.W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((unsigned*)(q + 4), __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((unsigned*)(q + 4) + 1, __float_as_uint(my), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fl, (unsigned)Fc | ((unsigned)Fw << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fl + 1, (ok ? 0u : 1u) | (flg & 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ok) { out[0 * cap + c] = W; out[1 * cap + c] = mx; out[2 * cap + c] = my; }
    }
    __syncthreads();
    for (int i = tid; i < S; i += PHD_T) {
        const float* r = srt + (size_t)i * 8;
        const int c = __float_as_int(r[6]);
        if (c < 0 || c >= K) continue;
        long long* q = acc + (size_t)c * 8;
        const unsigned* fl = (const unsigned*)(q + 5);
        const unsigned sc = g_ld32(fl), flg = g_ld32(fl + 1);
        if ((flg & 2u) && __float_as_int(((const float*)(q + 6))[3]) == i) continue;
        const float mean_x = __uint_as_float(g_ld32((const unsigned*)(q + 4))), mean_y = __uint_as_float(g_ld32((const unsigned*)(q + 4) + 1));
        bool ok = true;
        i64 qxx, qxy, qyy;
        fx_cov_terms((int)(sc & 0xFFFFu), mean_x, mean_y, r[0], r[1], r[2], r[3], r[4], r[5], qxx, qxy, qyy, ok);
        g_add64(q + 0, qxx); g_add64(q + 1, qxy); g_add64(q + 2, qyy);
        if (!ok) atomicOr((unsigned*)fl + 1, 1u);
    }
    __syncthreads();
    for (int c = tid; c < K; c += PHD_T) {
        long long* q = acc + (size_t)c * 8;
        const i64 cxx = g_ld64(q + 0), cxy = g_ld64(q + 1), cyy = g_ld64(q + 2), Wq = g_ld64(q + 3);
        const unsigned sc = g_ld32((const unsigned*)(q + 5)), flg = g_ld32((const unsigned*)(q + 5) + 1);
        const int Fc = (int)(sc & 0xFFFFu), Fw = (int)(sc >> 16);
        if (flg & 1u) {
            const float bad = __builtin_nanf("");
#pragma unroll
            for (int pl = 0; pl < 6; ++pl) out[pl * cap + c] = bad;
        } else {
            out[3 * cap + c] = fx_cov(cxx, Wq, Fc, Fw);
            out[4 * cap + c] = fx_cov(cxy, Wq, Fc, Fw);
            out[5 * cap + c] = fx_cov(cyy, Wq, Fc, Fw);
        }
    }
    __syncthreads();
    unsigned status = 0;
    int k_out = n_clusters < s_stop ? n_clusters : s_stop;
    if (k_out > cap) { k_out = cap; status |= PHD_STATUS_MAP_OVERFLOW; }
    int n_app = n_out0;
    if (k_out + n_app > cap) { n_app = cap - k_out; status |= PHD_STATUS_MAP_OVERFLOW; }
    const unsigned short* oidx = A.spill_out + (size_t)p * cap;
    for (int i = tid; i < n_app; i += PHD_T) {                          // untouched out-of-range features (src/phdfilter.cu:3311-3318)
        const int s = oidx[i];
        out[k_out + i] = in[s] * r_out_scale;
#pragma unroll
        for (int pl = 1; pl < 6; ++pl) out[pl * cap + k_out + i] = in[pl * cap + s];
    }
    if (tid == 0) {
        A.count_out[p] = k_out + n_app;
        if (rows_stride) ((int*)(out - 8))[6] = k_out + n_app;          // the export row's header (phd_export_kernel's layout)
        if (status) atomicOr(A.status, status);
        const int hm = k_out + n_out0;
        if (hm > __hip_atomic_load(A.max_map, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(A.max_map, hm);
    }
}

__global__ __launch_bounds__(PHD_T) void phd_merge_spill_kernel(UpdateArgs A)
{
    const int p = blockIdx.x;
    const int* meta = A.spill_meta + (size_t)p * 8;
    const int S = meta[0];
    if (S <= 0) return;                                                 // the LDS merge handled this particle
    const float r_scale = __int_as_float(meta[3]);                      // CPHD: the missed-detection factor of untouched features
    if (A.cfg.distanceMetric == 0) merge_spill_body<false>(A, p, S, meta[1], meta[2], r_scale, meta[4]);
    else merge_spill_body<true>(A, p, S, meta[1], meta[2], r_scale, meta[4]);
}

} // namespace phd
