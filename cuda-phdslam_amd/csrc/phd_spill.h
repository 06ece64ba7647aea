// phd_spill.h — the merge of a particle whose survivor list does not fit LDS (more than S_cap pruned update components:
// dense scans of large maps — the reference allows M = 256 measurements and has no cap on the map, src/phdfilter.cu:3390-3394,
// src/main.cpp:1003).  Part of the one translation unit phd_kernels.hip (device code, namespace phd).
//
// The update kernel writes such a particle's survivors to a record list in HBM (SpillRef, phd_lds.h) and leaves the merge
// to this kernel: the same greedy algorithm (seed = heaviest unmerged survivor, ties to the lowest slab index; absorb
// d < T; moment matching in weight order; stop at W == 0 — src/phdfilter.cu:2739-2890, src/gm_reduce.cpp:57-134) in its
// plain form, on global memory: rank by counting (keys tiled through LDS), one seed at a time, every thread testing its
// share of the unmerged survivors with the exact distance.  Correct for any list length up to spill_cap, slow by design:
// it is the path that replaces PHD_ERR_CAPACITY, not a fast path (LDS-resident lists take merge_in_lds).
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_merge.h"

namespace phd {

#define PHD_SPILL_TILE 2048

template <bool HELLINGER>
__device__ __forceinline__ void merge_spill_body(const UpdateArgs& A, int p, int S, int n_update, int n_out0, float r_out_scale, int src)
{
    __shared__ u64 s_keys[PHD_SPILL_TILE];
    __shared__ float s_seed[8];
    __shared__ int s_head, s_stop, s_nclusters;
    const int tid = threadIdx.x;
    const DevConfig& cfg = A.cfg;
    const float T = cfg.minSeparation;
    const int cap = A.cap;
    float* rec = A.spill_rec + (size_t)p * 2 * A.spill_cap * 8;        // arrival order
    float* srt = rec + (size_t)A.spill_cap * 8;                        // (weight desc, slab index asc) order
    int* assign = (int*)rec;                                           // reused once the sorted copy exists: one int per survivor
    const unsigned rows_stride = A.fuse_weights ? 0u : A.out_stride;
    float* out = A.map_out + (size_t)p * (rows_stride ? rows_stride : (size_t)6 * cap);
    // src: the slab the update kernel read (handed over in spill_meta: parent[p] has been reset to p by then)
    const float* in = A.map_in + (size_t)src * 6 * cap;

    // ---- rank by counting: rank_i = number of keys that sort before key_i (keys are unique) ----
    auto key_of = [&](int i) -> u64 {
        const int u0 = __float_as_int(rec[(size_t)i * 8 + 6]);
        const u32 ml = 0xFFFFFFFFu - (u32)(u0 >= NEAR_U_BASE ? u0 - NEAR_U_BASE + n_update : u0);
        return ((u64)orderable(rec[(size_t)i * 8 + 0]) << 32) | ml;
    };
    for (int i0 = 0; i0 < S; i0 += PHD_T) {
        const int i = i0 + tid;
        const u64 mine = i < S ? key_of(i) : 0ull;
        int rank = 0;
        for (int t0 = 0; t0 < S; t0 += PHD_SPILL_TILE) {
            const int tn = (S - t0 < PHD_SPILL_TILE) ? S - t0 : PHD_SPILL_TILE;
            __syncthreads();
            for (int j = tid; j < tn; j += PHD_T) s_keys[j] = key_of(t0 + j);
            __syncthreads();
            if (i < S)
                for (int j = 0; j < tn; ++j) rank += (s_keys[j] > mine) ? 1 : 0;
        }
        if (i < S) {
            const float* r = rec + (size_t)i * 8;
            float* d = srt + (size_t)rank * 8;
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = r[k];
        }
    }
    __syncthreads();                       // (block-scope visibility of the global writes: same CU, L1 write-through + barrier)
    __threadfence_block();
    for (int i = tid; i < S; i += PHD_T) assign[i] = -1;
    if (tid == 0) { s_head = 0; s_stop = 0x7FFFFFFF; s_nclusters = 0; }
    __syncthreads();

    // ---- greedy: one seed at a time ----
    for (;;) {
        if (tid == 0) {
            int h = s_head;
            while (h < S && assign[h] >= 0) ++h;
            s_head = h;
            if (h < S) {
                const float* r = srt + (size_t)h * 8;
                s_seed[0] = r[1]; s_seed[1] = r[2]; s_seed[2] = r[3]; s_seed[3] = r[4]; s_seed[4] = r[5];
                assign[h] = h;             // the seed heads its own cluster (membership of itself is decided by dself below)
                s_nclusters += 1;
            }
        }
        __syncthreads();
        const int h = s_head;
        if (h >= S) break;
        const float smx = s_seed[0], smy = s_seed[1], sxx = s_seed[2], sxy = s_seed[3], syy = s_seed[4];
        for (int i = h + 1 + tid; i < S; i += PHD_T) {
            if (assign[i] >= 0) continue;
            const float* r = srt + (size_t)i * 8;
            if (is_close<HELLINGER>(smx, smy, sxx, sxy, syy, r[1], r[2], r[3], r[4], r[5], T)) assign[i] = h;
        }
        __syncthreads();
        if (tid == 0) s_head = h + 1;
        __syncthreads();
    }
    const int n_clusters = s_nclusters;

    // ---- moment matching: one thread per cluster, members in sorted order (src/gm_reduce.cpp:103-118) ----
    // cluster index of seed h = number of seeds before h: every thread finds its c-th seed by a scan of the assignments
    for (int c0 = 0; c0 < n_clusters; c0 += PHD_T) {
#pragma clang fp contract(off)
        const int c = c0 + tid;
        if (c < n_clusters) {
            int h = -1, seen = 0;
            for (int i = 0; i < S; ++i)
                if (assign[i] == i) { if (seen == c) { h = i; break; } ++seen; }
            const float* sr = srt + (size_t)h * 8;
            const float smx = sr[1], smy = sr[2], sxx = sr[3], sxy = sr[4], syy = sr[5];
            const float dself = HELLINGER ? hellinger_dist(smx, smy, sxx, sxy, syy, smx, smy, sxx, sxy, syy)
                                          : mahal_dist(smx, smy, sxx, sxy, syy, smx, smy, sxx, sxy, syy);
            const bool selfok = dself < T;
            // exact, order-free sums (phd_fixsum.h), as in the LDS merges
            const int Fw = fx_field(sr[0]);
            FxSums fs = {0, 0, 0, 0, 0, 0, 0, 0, 0, true};
            for (int i = h; i < S; ++i) {
                if (assign[i] != h || (i == h && !selfok)) continue;
                const float* r = srt + (size_t)i * 8;
                fx_add_first(fs, Fw, smx, smy, r[0], r[1], r[2], r[3], r[4], r[5]);
            }
            int stop_at = 0x7FFFFFFF;
            if (fs.W == 0 && fs.ok && fs.ec < 255) stop_at = c;                       // src/phdfilter.cu:2821
            else if (!selfok) stop_at = c + 1;
            if (stop_at != 0x7FFFFFFF) atomicMin(&s_stop, stop_at);
            if (!(fs.W == 0 && fs.ok && fs.ec < 255) && c < cap) {
                float W, mx, my;
                fx_mean(fs, Fw, W, mx, my);
                const int Fc = fx_cov_anchor(Fw, fs.ec);
                bool ok = fs.ok && fs.ec < 255;
                for (int i = h; i < S; ++i) {
                    if (assign[i] != h || (i == h && !selfok)) continue;
                    const float* r = srt + (size_t)i * 8;
                    i64 qxx, qxy, qyy;
                    fx_cov_terms(Fc, mx, my, r[0], r[1], r[2], r[3], r[4], r[5], qxx, qxy, qyy, ok);
                    fs.cxx += qxx; fs.cxy += qxy; fs.cyy += qyy;
                }
                const float bad = __builtin_nanf("");
                out[0 * cap + c] = ok ? W : bad; out[1 * cap + c] = ok ? mx : bad; out[2 * cap + c] = ok ? my : bad;
                out[3 * cap + c] = ok ? fx_cov(fs.cxx, fs.W, Fc, Fw) : bad;
                out[4 * cap + c] = ok ? fx_cov(fs.cxy, fs.W, Fc, Fw) : bad;
                out[5 * cap + c] = ok ? fx_cov(fs.cyy, fs.W, Fc, Fw) : bad;
            }
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
