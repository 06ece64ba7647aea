// phd_merge.h — the greedy merge in LDS: exact closeness decision, single-shot merge (<= 256 survivors), merge rounds.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"
#include "phd_sort.h"

namespace phd {

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
    if (tid == 0) atomicMin((int*)&L.ctr[CTR_KOUT], n_clusters);   // the count is min(clusters, first stop): settled by the barrier below
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
    if (tid == 0) atomicMin((int*)&L.ctr[CTR_KOUT], n_clusters);   // as in merge_small
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
}

} // namespace phd
