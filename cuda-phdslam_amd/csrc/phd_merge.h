// phd_merge.h — the greedy merge in LDS: exact closeness decision, single-shot merge (<= 256 survivors), merge rounds.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"
#include "phd_sort.h"
#include "phd_fixsum.h"

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

// Is a seed within the merge distance of ITSELF (src/phdfilter.cu:2802-2806 evaluates d(s, s) like any other pair: a seed
// whose distance to itself is NaN — singular or non-finite covariance — stays out of its own cluster)?
// Mahalanobis, the common case without the four divisions: with a finite mean the offsets are exactly 0, and with
// |P| <= 2^60, det >= 2^-60 every quotient of mahal_dist() is finite (< 2^121), so the distance is exactly 0 (+-0 + +-0).
// Anything else evaluates the formula itself.
template <bool HELLINGER>
__device__ __forceinline__ bool seed_close_to_itself(float mx, float my, float xx, float xy, float yy, float T)
{
#pragma clang fp contract(off)
    if (!HELLINGER) {
        const float BIG = 1.152921504606846976e18f, SMALL = 8.67361737988403547e-19f;   // 2^60, 2^-60
        const float det = xx * yy - xy * xy;                     // = mahal_dist's det: (P + P) / 2 is P exactly below 2^60
        // (a NaN anywhere fails its comparison — fmaxf() would drop it)
        const bool tame = fabsf(xx) <= BIG && fabsf(xy) <= BIG && fabsf(yy) <= BIG && fabsf(mx) <= BIG && fabsf(my) <= BIG;
        if (tame && det >= SMALL) return 0.f < T;
    }
    const float dself = HELLINGER ? hellinger_dist(mx, my, xx, xy, yy, mx, my, xx, xy, yy) : mahal_dist(mx, my, xx, xy, yy, mx, my, xx, xy, yy);
    return dself < T;
}

// ------------------------------------------------------------------------------------------
// Moment matching by exact, order-free sums (phd_fixsum.h), shared by both LDS merges.
// The reference adds a cluster's members with a reduction tree (src/phdfilter.cu:2795-2881); round 2 grouped the
// survivors by seed with a second sort and walked each cluster sequentially.  Integer sums need neither: every survivor
// adds its terms to its cluster's accumulators with LDS atomics, in whatever order the lanes arrive.
//   A: weights, weighted means, the covariance scale (largest exponent)      -> per-cluster W, mean
//   B: weighted covariances about the merged mean                             -> per-cluster covariance
// recA[i] = (mean x, mean y, -, weight), recB[i] = (cov xx, xy, yy, assignment word) for the S survivors; the assignment word
// is (cluster index | index of the cluster's seed << 16), -1 for none: the seed's mean and weight — the anchors of the
// fixed-point scales — are read from the seed's own record, no per-cluster table.
// acc: 48 B per cluster for acc_clusters clusters at a time; more clusters than that take another sweep over the survivors
// (4096 x 256 x 64: ~320 clusters, room for 384).  Leaves the cluster count (min(n_clusters, first stop)) in ctr[CTR_KOUT].
// Every thread of the workgroup calls it.
// ------------------------------------------------------------------------------------------
template <bool HELLINGER, bool STAMPS>
__device__ __forceinline__ void moment_sums(const LDS_T(v4f)* recA, const LDS_T(v4f)* recB, int S, int n_clusters, int cap,
                                            LDS_T(long long)* acc, int acc_clusters, lds_i32 ctr, float T,
                                            float* __restrict__ out_slab, int tid, u64* st)
{
    const int Kall = n_clusters < cap ? n_clusters : cap;      // clusters past the map capacity are reported, not stored
    if (tid == 0) { ctr[CTR_NHEAD] = n_clusters; ctr[CTR_KOUT] = n_clusters; }
    for (int lo = 0; lo < Kall; lo += acc_clusters) {
    const int K = (Kall - lo) < acc_clusters ? (Kall - lo) : acc_clusters;
    // The accumulators: 48 B per cluster as SIX PLANES of K 8-byte words (sums: integers carried in doubles, phd_fixsum.h),
    //   pass A: W | x hi | x lo | y hi | y lo | (largest exponent, flags | seed's weight field << 8)
    //   pass B: cxx | cxy | cyy | W | (mean x, mean y) | (scales, flags)
    // The lanes of a wave add to the clusters of their survivors — arbitrary indices c: in a plane those fall on bank pair
    // c mod 32, all 32 in use; the record-per-cluster layout (48 B apart) left 16 of them, every atomic at least four-deep in
    // its banks — and LDS atomics are what this routine costs (the resident workgroups share the pipeline).
    // (the sums themselves are 64-bit INTEGERS — fx_i64() — added with ds_add_u64; plane 3 carries W as a double in pass B)
    LDS_T(double)* const accd = (LDS_T(double)*)acc;
    LDS_T(u32)* const acc32 = (LDS_T(u32)*)acc;
    LDS_T(i64)* const p0 = acc, * const p1 = acc + K, * const p2 = acc + 2 * K, * const p3 = acc + 3 * K, * const p4 = acc + 4 * K;
    LDS_T(double)* const p3d = accd + 3 * K;
    LDS_T(u32)* const w4 = acc32 + 8 * K;                       // plane 4 as words: (mean x, mean y) in pass B
    LDS_T(u32)* const w5 = acc32 + 10 * K;                      // plane 5 as words: (largest exponent | scales, flags)
    for (int t = tid; t < 3 * K; t += PHD_T) ((LDS_T(v4f)*)acc)[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#ifdef PHD_DUP_MOMENTS
    for (int dup_ = 0; dup_ <= PHD_DUP_MOMENTS; ++dup_)
#endif
    for (int i = tid; i < S; i += PHD_T) {
        const v4f a = recA[i], b = recB[i];
        const u32 aw = __float_as_uint(b.w);
        const int c = (int)(aw & 0xFFFFu) - lo;
        if (aw == 0xFFFFFFFFu || c < 0 || c >= K) continue;
        const int seed = (int)(aw >> 16);
        const v4f sa = recA[seed];                              // the seed's record: (mean x, mean y, -, weight)
        const int Fw = fx_field(sa.w);
        if (seed == i) {
            // the seed: leaves its weight field for the per-cluster steps, and is in its own cluster only if it is close to
            // itself (a NaN distance is not)
            const bool self = seed_close_to_itself<HELLINGER>(a.x, a.y, b.x, b.y, b.z, T);
            __hip_atomic_fetch_or(&w5[2 * c + 1], ((u32)Fw << 8) | (self ? 0u : 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!self) continue;
        }
        FxSumsD fs;
        fx_first_d(fs, Fw, sa.x, sa.y, a.w, a.x, a.y, b.x, b.y, b.z);
        __hip_atomic_fetch_add(p0 + c, fx_i64(fs.W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p1 + c, fx_i64(fs.xh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p2 + c, fx_i64(fs.xl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p3 + c, fx_i64(fs.yh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p4 + c, fx_i64(fs.yl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(&w5[2 * c], (u32)fs.ec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!fs.ok) __hip_atomic_fetch_or(&w5[2 * c + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    STAMP(8);
    // per cluster: W, mean (:2828), the stop rule (:2821); the planes are re-armed for pass B
    for (int c = tid; c < K; c += PHD_T) {
        FxSumsD fs;
        fs.W = (double)p0[c]; fs.xh = (double)p1[c]; fs.xl = (double)p2[c]; fs.yh = (double)p3[c]; fs.yl = (double)p4[c];   // (below 2^53: exact)
        const u32 ec = w5[2 * c], fl = w5[2 * c + 1];
        const bool ok = !(fl & 1u) && ec < 255u, selfok = !(fl & 2u);
        const int Fw = (int)((fl >> 8) & 0xFFu);
        // reference loop: W == 0 -> break (src/phdfilter.cu:2821); a seed left unmerged is re-picked and then yields W == 0
        int stop_at = 0x7FFFFFFF;
        if (fs.W == 0.0 && ok) stop_at = lo + c;
        else if (!selfok) stop_at = lo + c + 1;
        if (stop_at != 0x7FFFFFFF) atomicMin((int*)&ctr[CTR_KOUT], stop_at);
        float W, mx, my;
        fx_mean_d(fs, Fw, W, mx, my);
        const int Fc = fx_cov_anchor(Fw, (int)ec);
        p0[c] = 0; p1[c] = 0; p2[c] = 0;
        p3d[c] = fs.W;
        w4[2 * c] = __float_as_uint(mx); w4[2 * c + 1] = __float_as_uint(my);
        w5[2 * c] = (u32)Fc | ((u32)Fw << 16);
        w5[2 * c + 1] = (ok ? 0u : 1u) | (fl & 2u);
        if (ok) { out_slab[0 * cap + lo + c] = W; out_slab[1 * cap + lo + c] = mx; out_slab[2 * cap + lo + c] = my; }
    }
    __syncthreads();
    STAMP(9);
#ifdef PHD_DUP_MOMENTS
    for (int dup_ = 0; dup_ <= PHD_DUP_MOMENTS; ++dup_)
#endif
    for (int i = tid; i < S; i += PHD_T) {
        const v4f a = recA[i], b = recB[i];
        const u32 aw = __float_as_uint(b.w);
        const int c = (int)(aw & 0xFFFFu) - lo;
        if (aw == 0xFFFFFFFFu || c < 0 || c >= K) continue;
        const v2f hm = ((LDS_T(v2f)*)w4)[c];                    // (mean x, mean y)
        const u32 sc = w5[2 * c], hfl = w5[2 * c + 1];          // (scales, flags)
        // (flag 2: the seed is not in its own cluster, decided in pass A)
        if ((hfl & 2u) && (int)(aw >> 16) == i) continue;
        bool ok = true;
        double qxx, qxy, qyy;
        fx_cov_terms_d((int)(sc & 0xFFFFu), hm.x, hm.y, a.w, a.x, a.y, b.x, b.y, b.z, qxx, qxy, qyy, ok);
        __hip_atomic_fetch_add(p0 + c, fx_i64(qxx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p1 + c, fx_i64(qxy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p2 + c, fx_i64(qyy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!ok) __hip_atomic_fetch_or(&w5[2 * c + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    // per cluster: covariance = sum / W (:2879; the symmetric form needs no force_symmetric_covariance)
    for (int c = tid; c < K; c += PHD_T) {
        const double cxx = (double)p0[c], cxy = (double)p1[c], cyy = (double)p2[c], Wq = p3d[c];
        const u32 sc = w5[2 * c];
        const int Fc = (int)(sc & 0xFFFFu), Fw = (int)(sc >> 16);
        if (w5[2 * c + 1] & 1u) {
            const float bad = __builtin_nanf("");
#pragma unroll
            for (int pl = 0; pl < 6; ++pl) out_slab[pl * cap + lo + c] = bad;
        } else {
            out_slab[3 * cap + lo + c] = fx_cov_d(cxx, Wq, Fc, Fw);
            out_slab[4 * cap + lo + c] = fx_cov_d(cxy, Wq, Fc, Fw);
            out_slab[5 * cap + lo + c] = fx_cov_d(cyy, Wq, Fc, Fw);
        }
    }
    if (lo + acc_clusters < Kall) __syncthreads();             // (uniform) the next sweep re-arms the planes these loops read
    }
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
    lds_u32 plist = L.splist;                    // candidate pairs (k << 16 | l): survivor planes, dead since the sorted staging
    const int pcap = L.splist_cap;
    // ---- closeness rows.  Work units are HALF waves: 32 consecutive positions (half-block hb) x 16 columns (column unit
    //      cu, columns 16 cu .. 16 cu + 15, only units with a column below the half-block's last row), two units per
    //      wave iteration, dealt round-robin to the waves; a unit fills the 16-bit quarter cu of its rows' words.
    //      (With 64-position units a 148-survivor mixture paid for 24 units of which 44 % was real work; now 15.)
    const int ncu = (S + 15) >> 4, nhb = (S + 31) >> 5;
    int n_half = 0;
    for (int hb = 0; hb < nhb; ++hb) n_half += (2 * hb + 2 < ncu) ? 2 * hb + 2 : ncu;
    {
        static_assert(PHD_SMALL_S == 256, "merge_small uses four 64-bit words per row");
        // (the rows are not cleared: every quarter a later phase reads — words lc <= k / 64 of the rows k < S — is written below)
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
    // ---- membership: every position joins the first seed of its row (a seed joins itself); its cluster index — the seeds
    //      before its owner, i.e. the output position — goes into the record, the seed leaves the cluster's record
    const u64 s0 = L.sseed[0], s1 = L.sseed[1], s2 = L.sseed[2], s3 = L.sseed[3];
    // (the rows and the seed masks are dead after this pass: the sums' accumulators take their place, 12 KB for 256 clusters)
    LDS_T(long long)* const acc = (LDS_T(long long)*)L.srow;
    int owner = 0;
    bool is_seed = false;
    if (tid < S) {
        const int k = tid;
        const u64 sw = (k < 64) ? s0 : (k < 128) ? s1 : (k < 192) ? s2 : s3;
        is_seed = (sw >> (k & 63)) & 1ull;
        owner = k;
        if (!is_seed) {
            const int kbk = k >> 6;                  // words beyond the position's own block were never written
            const u64 m0 = L.srow[k * 4 + 0] & s0, m1 = (kbk >= 1) ? (L.srow[k * 4 + 1] & s1) : 0ull,
                      m2 = (kbk >= 2) ? (L.srow[k * 4 + 2] & s2) : 0ull, m3 = (kbk >= 3) ? (L.srow[k * 4 + 3] & s3) : 0ull;
            owner = m0 ? __builtin_ctzll(m0) : m1 ? 64 + __builtin_ctzll(m1) : m2 ? 128 + __builtin_ctzll(m2)
                                                                            : 192 + __builtin_ctzll(m3);
        }
    }
    __syncthreads();                                  // every row has been read: the accumulators may overwrite them
    if (tid < S) {
        const int k = tid, o = owner;
        const u64 below = (o & 63) ? (~0ull >> (64 - (o & 63))) : 0ull;
        int c = 0;                                    // cluster index = seeds before the owner
        c += (o >= 64) ? __popcll(s0) : __popcll(s0 & below);
        if (o >= 64) c += (o >= 128) ? __popcll(s1) : __popcll(s1 & below);
        if (o >= 128) c += (o >= 192) ? __popcll(s2) : __popcll(s2 & below);
        if (o >= 192) c += __popcll(s3 & below);
        ((LDS_T(int)*)&L.sB[k])[3] = c | (o << 16);   // the assignment word: cluster | its seed's position
    }
    if (STAMPS && tid == 0) {
        const u64 tq3 = __builtin_amdgcn_s_memrealtime();
        st[12] += tq1 - tq0; st[13] += tq2 - tq1; st[14] += tq3 - tq2; st[15] += 1;
    }
    STAMP(7);
    // ---- moment matching by exact, order-free sums (moment_sums above)
    const int n_clusters = __popcll(s0) + __popcll(s1) + __popcll(s2) + __popcll(s3);
    moment_sums<HELLINGER, STAMPS>(L.sA, L.sB, S, n_clusters, cap, acc, PHD_SMALL_S, L.ctr, T, out_slab, tid, st);
    __syncthreads();
    STAMP(10);
}

// ------------------------------------------------------------------------------------------
// merge_tail: ONE-SHOT finish of the rounds (round 6).
//
// The rounds resolve 64 candidate seeds at a time, four barriers each, and their LAST rounds are nearly empty: at
// 4096 x 256 x 64 the list of unmerged survivors reads ~880, ~300, ~150, ~55, (~5) at the start of the 4.25 rounds
// (tools/round_tail.py), so rounds 3, 4 and 5 pay their fixed cost — window matrix, seed resolution by one wave, seed records,
// ordered compaction, four barriers — for one or two busy waves.  Once at most PHD_TAIL_N survivors are listed the rest of the
// greedy merge is finished in one shot with merge_small's structure on the listed survivors (in list order = weight order):
//   rows     row_k = {l < k : close(k, l)} for ALL pairs of listed survivors: far-pair filter in sign-bit form (the rounds' own:
//            E_k + E_l - 2 m_k.m_l < 0, gA.z) — lane = position k of a 64-row block, a unit = 16 columns whose terms reach the
//            scalar registers through v_readlane — then the exact decision on the listed pairs, one pair per thread;
//   seeds    s_k = not exists l < k : close(k, l) and s_l, by one wave, 64 positions at a time (earlier blocks are final);
//   members  every position joins the first seed of its row; cluster index = kbase + seeds before its owner.
// Exactly the decisions the rounds would take (the same greedy, the same exact test), five barriers instead of ~2.3 rounds x 4.
// LDS: the bytes behind the two round lists up to the end of the update's small arrays are dead by now (window, seed records,
// matrix parts, measurement arrays): rows [N][4] u64 | tA, tB [N] float4 (the listed survivors' records by position) | seed
// masks | candidate pairs.  tail_cap = the N this filter's layout has room for (0: never).
// ------------------------------------------------------------------------------------------
#ifndef PHD_TAIL_N
#define PHD_TAIL_N 192
#endif
// rows [N][N / 64] u64 | tA, tB [N] float4 | seed masks | eight wave-private pair lists of 64 entries
__host__ __device__ constexpr int merge_tail_bytes(int N) { return N * (N / 64) * 8 + 2 * N * 16 + 64 + PHD_NW * 64 * 4; }

template <bool HELLINGER, bool STAMPS>
__device__ __forceinline__ int merge_tail(const Lds& L, lds_u16 cur, int n, int cap_n, int kbase, float T, int tid)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = cap_n >> 6;                                          // 64-bit words per row (3 or 2)
    LDS_T(u64)* const rows = (LDS_T(u64)*)L.rwin;                       // [cap_n][rw]: row_k = {l < k : close(k, l)}
    LDS_T(u32)* const rows32 = (LDS_T(u32)*)rows;
    LDS_T(v4f)* const tA = (LDS_T(v4f)*)(rows + rw * cap_n);            // [cap_n] (mean x, mean y, E, weight)
    LDS_T(v4f)* const tB = tA + cap_n;                                  // [cap_n] (cov xx, xy, yy, -)
    LDS_T(u64)* const sseed = (LDS_T(u64)*)(tB + cap_n);                // [4] (+ 4 spare)
    lds_u32 wlist = (lds_u32)(sseed + 8) + wave * 64;                   // this wave's waiting pairs (k << 16 | l)
    LDS_T(v4f)* const gB = L.gB;
    // (the listed survivors' records are staged by position in tA / tB and the rows are cleared by the compaction of the round before —
    //  it knows the list's length before it writes, and writes these arrays INSTEAD of a next window; its closing barrier is ours: +0.7 %)
    // ---- closeness rows: units of 64 positions x 16 columns, only units with a column below the block's last row.
    //      The filter marks candidate pairs; the wave collects its units' marked pairs in a wave-private list (in order, no
    //      barrier) and takes the exact decisions ONE PAIR PER LANE when the next unit's pairs would not fit (and at the
    //      end): a pair that is close sets its bit in the cleared row with an LDS atomic OR (a word's bits come from several waves).
    const int nb = (n + 63) >> 6, ncu = (n + 15) >> 4;
    int n_units = 0;
    for (int b = 0; b < nb; ++b) n_units += (4 * b + 4 < ncu) ? 4 * b + 4 : ncu;
    int waiting = 0;                                                 // (uniform)
    for (int u0 = wave; u0 < n_units; u0 += PHD_NW) {
        int b = 0, cu = u0;
        for (;;) { const int c = (4 * b + 4 < ncu) ? 4 * b + 4 : ncu; if (cu < c) break; cu -= c; ++b; }
        const int lbase = 16 * cu;
        const int k = 64 * b + lane;
        const bool kvalid = k < n;
        const v4f ka = tA[kvalid ? k : n - 1];
        const v4f ca = tA[(lbase + (lane & 15)) < n ? lbase + (lane & 15) : n - 1];   // lane j < 16: column lbase + j
        const float cE = ca.z, c2x = -2.f * ca.x, c2y = -2.f * ca.y;
        const v2f kE2 = (v2f){ka.z, ka.z}, kx2 = (v2f){ka.x, ka.x}, ky2 = (v2f){ka.y, ka.y};
        u32 cand = 0;
#pragma unroll
        for (int c = 14; c >= 0; c -= 2) {                       // descending: v_alignbit shifts the mask left
            v2f t = (v2f){lane_f(cE, c), lane_f(cE, c + 1)} + kE2;
            t = __builtin_elementwise_fma((v2f){lane_f(c2x, c), lane_f(c2x, c + 1)}, kx2, t);
            t = __builtin_elementwise_fma((v2f){lane_f(c2y, c), lane_f(c2y, c + 1)}, ky2, t);
            cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(t.y), 31);
            cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(t.x), 31);
        }
        // only earlier positions count (l < k), and only rows of listed survivors
        const int nlt = k - lbase;
        cand &= (!kvalid || nlt <= 0) ? 0u : (nlt >= 16 ? 0xFFFFu : ((1u << nlt) - 1u));
        if (__ballot(cand != 0u) == 0ull) continue;              // uniform
        const int np = __popc(cand);
        const int incl = (int)wave_incl_scan((u32)np);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (waiting + tot > 64 && waiting) {                     // uniform: decide what is waiting first
            if (lane < waiting) {
                const u32 pr = wlist[lane];
                const int pk = (int)(pr >> 16), pl = (int)(pr & 0xFFFFu);
                const v4f qa = tA[pk], qb = tB[pk], la = tA[pl], lb = tB[pl];
                if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, qa.x, qa.y, qb.x, qb.y, qb.z, T))
                    __hip_atomic_fetch_or(rows32 + pk * 2 * rw + (pl >> 5), 1u << (pl & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            waiting = 0;
        }
        if (tot <= 64) {
            int pos = waiting + incl - np;
            while (cand) {
                const int j = __builtin_ctz(cand);
                cand &= cand - 1;
                wlist[pos++] = ((u32)k << 16) | (u32)(lbase + j);
            }
            waiting += tot;
        } else {
            // more than 64 marked pairs in one unit (the Hellinger metric has no cheap filter): per lane on its marked columns
            u32 bits = 0;
            if (cand) {
                const v4f kb = tB[k];
                while (cand) {
                    const int j = __builtin_ctz(cand);
                    cand &= cand - 1;
                    const v4f la = tA[lbase + j], lb = tB[lbase + j];
                    if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, ka.x, ka.y, kb.x, kb.y, kb.z, T)) bits |= 1u << j;
                }
                if (bits) __hip_atomic_fetch_or(rows32 + k * 2 * rw + (lbase >> 5), bits << (lbase & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (waiting) {
        if (lane < waiting) {
            const u32 pr = wlist[lane];
            const int pk = (int)(pr >> 16), pl = (int)(pr & 0xFFFFu);
            const v4f qa = tA[pk], qb = tB[pk], la = tA[pl], lb = tB[pl];
            if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, qa.x, qa.y, qb.x, qb.y, qb.z, T))
                __hip_atomic_fetch_or(rows32 + pk * 2 * rw + (pl >> 5), 1u << (pl & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    // ---- seeds: one wave, block by block (merge_small's resolution): earlier blocks are final, inside a block the ballot fixed point
    if (wave == 0) {
        u64 sd[3] = {0ull, 0ull, 0ull};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b < nb) {
                const int k = 64 * b + lane;
                const bool kvalid = k < n;
                const int kk = kvalid ? k : n - 1;
                const u64 r0 = rows[kk * rw + 0], r1 = (b >= 1) ? rows[kk * rw + 1] : 0ull, r2 = (b >= 2) ? rows[kk * rw + 2] : 0ull;
                bool blocked = false;
                if (b > 0) blocked = blocked || ((r0 & sd[0]) != 0ull);
                if (b > 1) blocked = blocked || ((r1 & sd[1]) != 0ull);
                const u64 row = (b == 0) ? r0 : (b == 1) ? r1 : r2;   // within the block
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
        if (lane < 3) sseed[lane] = (lane == 0) ? sd[0] : (lane == 1) ? sd[1] : sd[2];
    }
    __syncthreads();
    // ---- membership: every position joins the first seed of its row (a seed joins itself); the assignment word of the
    //      survivor: cluster index (seeds of the rounds + seeds before its owner) | its seed's survivor index
    const u64 s0 = sseed[0], s1 = sseed[1], s2 = sseed[2];
    if (tid < n) {
        const int k = tid, kbk = k >> 6;              // (words beyond the position's own block are zero: cleared, never set)
        const u64 sw = (kbk == 0) ? s0 : (kbk == 1) ? s1 : s2;
        int o = k;
        if (!((sw >> (k & 63)) & 1ull)) {
            const u64 m0 = rows[k * rw + 0] & s0, m1 = (kbk >= 1) ? (rows[k * rw + 1] & s1) : 0ull, m2 = (kbk >= 2) ? (rows[k * rw + 2] & s2) : 0ull;
            o = m0 ? __builtin_ctzll(m0) : m1 ? 64 + __builtin_ctzll(m1) : 128 + __builtin_ctzll(m2);
        }
        const u64 below = (o & 63) ? (~0ull >> (64 - (o & 63))) : 0ull;
        int c = 0;                                    // seeds before the owner
        c += (o >= 64) ? __popcll(s0) : __popcll(s0 & below);
        if (o >= 64) c += (o >= 128) ? __popcll(s1) : __popcll(s1 & below);
        if (o >= 128) c += __popcll(s2 & below);
        ((LDS_T(int)*)gB)[4 * cur[k] + 3] = (kbase + c) | ((int)cur[o] << 16);
    }
    __syncthreads();                                  // the moment sums' accumulators overwrite the lists and these arrays
    return __popcll(s0) + __popcll(s1) + __popcll(s2);
}

// ------------------------------------------------------------------------------------------
// the greedy merge on the survivors held in LDS; writes the merged map to the output slab.
// Returns (in ctr[CTR_KOUT]) the number of merged Gaussians.
// ------------------------------------------------------------------------------------------
template <bool HELLINGER, bool STAMPS>
__device__ __forceinline__ void merge_in_lds(const Lds& L, int S_cap, int n_surv, const DevConfig& cfg, float* __restrict__ out_slab,
                             int cap, int tid, u64* st, int n_update, bool packed, const BucketMap& bm)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: uniform, kept in an SGPR
    const float T = cfg.minSeparation;
    // guard band of the trace filter; T <= 0: "2 d2 >= -(..)" always holds -> far, like the exact test
    const float Tpre = (T > 0.f) ? T * 1.01f : -1.f;
    const int S = n_surv;
    if (tid == 0) { L.ctr[CTR_KOUT] = 0; L.ctr[CTR_NHEAD] = 0; }
    if (S == 0) { __syncthreads(); return; }
    if (S <= PHD_SMALL_S) { merge_small<HELLINGER, STAMPS>(L, S_cap, S, cfg, out_slab, cap, tid, st, n_update); return; }

    // ---- sort 1: (weight desc, slab index asc); leaves the survivors as the float4 arrays gA / gB ----
    int n_pad = 2;
    while (n_pad < S) n_pad <<= 1;
    if (bucket_sort_survivors(L, S, S_cap, tid, lane, wave, n_update, Tpre, HELLINGER, bm)) { /* counting sort did it */ }
    else {
        if (n_pad <= PHD_T) rank_sort_survivors(L, S, tid, n_update);
        else if (n_pad <= 2 * PHD_T) sort_survivors<2>(L, S, n_pad, tid, n_update, packed);
        else if (n_pad <= 4 * PHD_T) sort_survivors<4>(L, S, n_pad, tid, n_update, packed);
        else sort_survivors<8>(L, S, n_pad, tid, n_update, packed);
        planes_to_aos(L, S, tid, Tpre, HELLINGER);
    }
    STAMP(6);

    // ---- rounds: 64 live candidates at a time ---------------------------------------------------
    // `cur` lists the still-unmerged survivors in (weight desc) order.  Per round the first 64 of
    // them form the window: their pairwise closeness decides which are seeds (a candidate is a seed
    // iff no earlier seed is close to it), every later survivor is assigned to the first seed it is
    // close to, and the list is compacted.  Exactly the reference's greedy loop, 64 seeds at a time.
    //
    // Far-pair filter (filter_term(), phd_sort.h): pair (a, b) is a candidate iff E_a + E_b - 2 ma.mb < 0.  One side
    // of every test is wave-uniform (a window column / a seed) and sits in scalar registers, so a pair costs one
    // packed add, two packed FMAs for TWO pairs, and one v_alignbit per pair that shifts the sign bit into the mask.
    LDS_T(v4f)* const gA = L.gA;
    LDS_T(v4f)* const gB = L.gB;
    LDS_T(int)* const asg = (LDS_T(int)*)L.gB;           // assignment word of survivor i (cluster | seed's index << 16): asg[4 i + 3]
    lds_u16 ul_a = L.ulist, ul_b = ul_a + S_cap;         // two u16 lists where the sort's keys were
    lds_i32 wpos = (lds_i32)L.rwin;
    LDS_T(v4f)* const wA = (LDS_T(v4f)*)(L.rwin + 64);         // the window's copy of gA / gB
    LDS_T(v4f)* const wB = (LDS_T(v4f)*)(L.rwin + 64 + 256);
    // the round's seeds in order, laid out for the packed filter — a register PAIR per operand, no moves to assemble it:
    //   sP[k] = (E_2k, E_2k+1, -2 mx_2k, -2 mx_2k+1)   sQ[k] = (-2 my_2k, -2 my_2k+1)   sG[r] = (cov xx, xy, yy, cluster index)
    LDS_T(v4f)* const sP = (LDS_T(v4f)*)(L.rwin + 64 + 512);         // 36 entries (64 seeds + padding to a multiple of 8)
    LDS_T(v2f)* const sQ = (LDS_T(v2f)*)(sP + 36);                   // 36 entries
    LDS_T(v4f)* const sG = sP + 72;
    LDS_T(float)* const sPf = (LDS_T(float)*)sP;
    LDS_T(float)* const sQf = (LDS_T(float)*)sQ;
#ifdef PHD_PRICE_WAVE_PASS
    // PRICING EXPERIMENT (round 6; tools/ab_bench.sh): what ONE pass of a wave-local sequential greedy costs
    // inside this kernel, at its residency, beside its other workgroups — the seed is the first unmerged lane of the wave's
    // 64-survivor block (the block is in weight order), its record reaches the scalar registers through v_readlane, every
    // unmerged lane takes the exact decision against it, a ballot retires the members, the members store an assignment word.
    // The block stays in REGISTERS (the cheapest form: a component of more than 64 members would re-read LDS every pass), no
    // barrier, no LDS read inside a pass.  PHD_PRICE_WAVE_PASS passes per wave; the stores land in assignment words that the
    // rounds below overwrite, so the results are unchanged.
    {
        const int pi = wave * 64 + lane;
        const bool pvalid = pi < S;
        const v4f pa = gA[pvalid ? pi : 0], pb = gB[pvalid ? pi : 0];
        const u64 all = __ballot(pvalid);
        u64 unmerged = all;
        int pcnt = 0;
        if (all)
        for (int k = 0; k < PHD_PRICE_WAVE_PASS; ++k) {
            if (!unmerged) unmerged = all;                       // (re-armed: every pass does the work of a live component)
            const int sd = __builtin_ctzll(unmerged);
            const float smx = lane_f(pa.x, sd), smy = lane_f(pa.y, sd), sxx = lane_f(pb.x, sd), sxy = lane_f(pb.y, sd), syy = lane_f(pb.z, sd);
            const bool mine = (unmerged >> lane) & 1ull;
            const bool cl = mine && is_close<HELLINGER>(smx, smy, sxx, sxy, syy, pa.x, pa.y, pb.x, pb.y, pb.z, T);
            const u64 mm = __ballot(cl) | (1ull << sd);
            if (cl) asg[4 * pi + 3] = pcnt | ((wave * 64 + sd) << 16);
            unmerged &= ~mm;
            ++pcnt;
            asm volatile("" ::: "memory");
        }
    }
#endif
    for (int i = tid; i < S; i += PHD_T) ul_a[i] = (u16)i;
    int n_u = S;
    int kbase = 0;                                             // seeds of the rounds so far = clusters opened
    lds_u16 cur = ul_a, nxt = ul_b;
    // the first window: broadcast-friendly copy of the first 64 survivors (later windows are written by the compaction
    // of the round before: the first 64 entries it keeps ARE the next window)
    if (tid < 64) {
        const int i = tid < S ? tid : S - 1;
        wpos[tid] = i;
        v4f a = gA[i];
        if (tid >= S) a.z = INFINITY;                      // padding: never a candidate
        wA[tid] = a;
        wB[tid] = gB[i];
    }
    __syncthreads();
    // the one-shot finish (merge_tail above): what this filter's layout has room for behind the round lists
    int tail_cap = 0;
#ifndef PHD_NO_TAIL
    {
        static_assert(PHD_TAIL_N == 192 || PHD_TAIL_N == 128, "merge_tail holds rows of two or three 64-bit words");
        const int room = (int)((LDS_T(unsigned char)*)L.out_idx - (LDS_T(unsigned char)*)L.rwin);
        tail_cap = room >= merge_tail_bytes(PHD_TAIL_N) ? PHD_TAIL_N : room >= merge_tail_bytes(128) ? 128 : 0;
    }
#endif
    while (n_u > 0) {
        if (n_u <= tail_cap && n_u > 64) {                      // (uniform; a list of at most 64 is one more window round: cheaper)
            u64 tt0 = 0;
            if (STAMPS && tid == 0) tt0 = __builtin_amdgcn_s_memrealtime();
            kbase += merge_tail<HELLINGER, STAMPS>(L, cur, n_u, tail_cap, kbase, T, tid);
            if (STAMPS && tid == 0) st[31] += __builtin_amdgcn_s_memrealtime() - tt0;
            n_u = 0;
            break;
        }
        u64 tq0 = 0, tq1 = 0, tq2 = 0;
        if (STAMPS && tid == 0) tq0 = __builtin_amdgcn_s_memrealtime();
        const int nwin = n_u < 64 ? n_u : 64;
        const int nrest = n_u - nwin;
        u64 tf[6] = {0, 0, 0, 0, 0, 0};
#define RSTAMP(k) do { if (STAMPS && tid == 0) tf[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
        RSTAMP(0);
        // every wave holds the window in registers, lane l <-> candidate l; a column's terms reach the scalar
        // registers through v_readlane
        const v4f ka = wA[lane];
        const float kx = ka.x, ky = ka.y, kE = ka.z;
        const float m2x = -2.f * kx, m2y = -2.f * ky;
        // (1) closeness matrix rows: lane = candidate k, wave = column block [COLS*wave, COLS*(wave+1)).
        //     The filter marks candidate columns; the exact test runs on the marked bits only.
#ifdef PHD_DUP_MATRIX
        for (int dup_ = 0; dup_ <= PHD_DUP_MATRIX; ++dup_)
#endif
        {
#ifdef PHD_DUP_MATRIX
            asm volatile("" ::: "memory");
#endif
            const int k = lane;
            u32 cand = 0;
            const v2f kE2 = (v2f){kE, kE}, kx2 = (v2f){kx, kx}, ky2 = (v2f){ky, ky};
#pragma unroll
            for (int c = PHD_COLS - 2; c >= 0; c -= 2) {      // descending: v_alignbit shifts the mask left
                const int l = wave * PHD_COLS + c;
                v2f t = (v2f){lane_f(kE, l), lane_f(kE, l + 1)} + kE2;
                t = __builtin_elementwise_fma((v2f){lane_f(m2x, l), lane_f(m2x, l + 1)}, kx2, t);
                t = __builtin_elementwise_fma((v2f){lane_f(m2y, l), lane_f(m2y, l + 1)}, ky2, t);
                cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(t.y), 31);
                cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(t.x), 31);
            }
            // only earlier candidates count (l < k), and only rows of real candidates
            const int nlt = k - wave * PHD_COLS;
            cand &= (k >= nwin || nlt <= 0) ? 0u : (nlt >= PHD_COLS ? ((1u << PHD_COLS) - 1u) : ((1u << nlt) - 1u));
            RSTAMP(1);
            // The marked pairs are few (a wave's 64 rows x 8 columns hold ~20) and unevenly spread over the rows: a loop per row
            // runs as long as the fullest row with a handful of lanes active.  So the wave lists its pairs (wave-private LDS,
            // in-order — no barrier) and takes the exact decisions ONE PAIR PER LANE; the results are OR-ed into the rows.
            const u64 anyc = __ballot(cand != 0u);
            LDS_T(u32)* const myrow = &L.part[wave * 64 + k];
            int tot = 0, incl = 0;
            const int np = __popc(cand);
            if (anyc) {                                             // uniform
                incl = (int)wave_incl_scan((u32)np);
                tot = __builtin_amdgcn_readlane(incl, 63);
            }
            if (anyc && tot <= 64) {                                // uniform: the usual case
                LDS_T(u16)* const plist = (LDS_T(u16)*)L.win + wave * 64;     // (`win` is idle during the merge)
                int pos = incl - np;
                while (cand) {
                    const int c = __builtin_ctz(cand);
                    cand &= cand - 1;
                    plist[pos++] = (u16)((k << 3) | c);
                }
                *myrow = 0u;
                if (lane < tot) {
                    const u32 pr = plist[lane];
                    const int kk = (int)(pr >> 3), c = (int)(pr & 7u), l = wave * PHD_COLS + c;
                    const v4f pa = wA[kk], pb = wB[kk], la = wA[l], lb = wB[l];
                    if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, pa.x, pa.y, pb.x, pb.y, pb.z, T))
                        __hip_atomic_fetch_or(&L.part[wave * 64 + kk], 1u << c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                u32 bits = 0;
                if (cand) {
                    const v4f kb = wB[k];
                    while (cand) {
                        const int c = __builtin_ctz(cand);
                        cand &= cand - 1;
                        const int l = wave * PHD_COLS + c;
                        const v4f la = wA[l], lb = wB[l];
                        if (is_close<HELLINGER>(la.x, la.y, lb.x, lb.y, lb.z, kx, ky, kb.x, kb.y, kb.z, T)) bits |= (1u << c);
                    }
                }
                *myrow = bits;
            }
        }
        __syncthreads();
        if (STAMPS && tid == 0) tq1 = __builtin_amdgcn_s_memrealtime();
        // (2) seeds: s_k = not exists l < k : close(k,l) and s_l.  The recursion is well founded, so the
        //     parallel iteration s <- F(s) reaches its unique fixed point (position k is final after k+1
        //     sweeps; in practice a handful).  ONE wave resolves it and writes the round's seed records (3); the others wait at a
        //     barrier and read the mask.  (Rounds 1-3 had every wave compute the same mask and write the same records to save
        //     that barrier — right while the kernel was latency-bound; with three workgroups per CU the vector ALUs issue 89 % of
        //     the cycles and seven redundant copies of ~70 instructions per round cost more than the barrier: +1.5 %.)
        u64 seeds = 0ull;
        if (wave == 0) {
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
                // the cluster of a window candidate: its own if it is a seed, else the first seed it is close to.  Cluster
                // index = seeds before it, over all rounds = its position in the output (descending seed weight)
                const int owner = ((seeds >> lane) & 1ull) ? lane : __builtin_ctzll(row & seeds);
                asg[4 * wpos[lane] + 3] = (kbase + __popcll(seeds & ((1ull << owner) - 1ull))) | (wpos[owner] << 16);
            }
        }
        if (STAMPS && tid == 0) tq2 = __builtin_amdgcn_s_memrealtime();
        // (3) the survivors after the window.  The round's SEEDS (typically 40 of the 64 candidates) are listed in
        //     order as records (-2 mx, -2 my, E, window index) by wave 0.  The listed survivors are dealt to the waves in blocks of 64,
        //     one survivor per lane: the lane tests its survivor against ALL seeds (records by LDS broadcast, two seeds
        //     per packed operation, the sign bit of E_s + E_e - 2 ms.me shifted into a 64-bit mask whose ascending bits
        //     are ascending seed order) and, in the same pass, takes the exact decision on the marked seeds until the
        //     first hit.  One trip through LDS per survivor and round.
        if (wave == 0) {
            if (lane == 0) ((LDS_T(u64)*)L.red)[0] = seeds;    // (the reduction scratch is idle during the merge)
            const int nseeds = __popcll(seeds);
            const int rank = __popcll(seeds & lanemask_lt());
            if ((seeds >> lane) & 1ull) {
                const int pr = 4 * (rank >> 1) + (rank & 1);
                sPf[pr] = kE; sPf[pr + 2] = m2x; sQf[rank] = m2y;
                v4f g = wB[lane];
                g.w = __int_as_float((kbase + rank) | (wpos[lane] << 16));   // what a survivor that joins this seed stores
                sG[rank] = g;
            }
            if (lane < 8) {                                       // pad to a multiple of 8: never a candidate
                const int rp = nseeds + lane, pr = 4 * (rp >> 1) + (rp & 1);
                sPf[pr] = INFINITY; sPf[pr + 2] = 0.f; sQf[rp] = 0.f;
            }
        }
        __syncthreads();
        {
            const u64 sv = ((LDS_T(u64)*)L.red)[0];
            seeds = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)sv);
        }
        const int nseeds = __popcll(seeds);
        const int per = (nrest + PHD_T - 1) / PHD_T;           // entries per thread
        RSTAMP(2);
        kbase += nseeds;                                        // (uniform: every wave read the same seed mask)
        int kept = 0;
        u32 keepbits = 0;                                       // bit q: this thread's q-th entry stays listed
#ifdef PHD_DUP_ASSIGN
        for (int dup_ = 0; dup_ <= PHD_DUP_ASSIGN; ++dup_) { kept = 0; keepbits = 0; asm volatile("" ::: "memory");
#endif
        for (int q = 0; q < per; ++q) {
            // thread t owns the contiguous entries [t per, (t + 1) per) of the list (what the ordered compaction needs)
            const int e = tid * per + q;
            const bool ev = e < nrest;
            if (__ballot(ev) == 0ull) break;                    // uniform: nothing left for this wave
            const int i = ev ? cur[64 + e] : 0;
            v4f ea = gA[i];
            const v4f fb = gB[i];
            if (!ev) ea.z = INFINITY;
            const v2f eE2 = (v2f){ea.z, ea.z}, ex2 = (v2f){ea.x, ea.x}, ey2 = (v2f){ea.y, ea.y};
            u32 mlo = 0, mhi = 0;
            // eight seeds per trip, their records requested together (LDS broadcast reads), descending: v_alignbit
            // shifts the mask left
#define PHD_SEED_PAIR(M, p_, q_) do {                                                                  \
                v2f t_ = (p_).xy + eE2;                                                                  \
                t_ = __builtin_elementwise_fma((p_).zw, ex2, t_);                                        \
                t_ = __builtin_elementwise_fma((q_), ey2, t_);                                           \
                M = __builtin_amdgcn_alignbit(M, __float_as_uint(t_.y), 31);                             \
                M = __builtin_amdgcn_alignbit(M, __float_as_uint(t_.x), 31); } while (0)
#define PHD_SEED_OCTET(M, r) do {                                                                      \
                const int k_ = (r) >> 1;                                                                 \
                const v4f p0 = sP[k_ + 0], p1 = sP[k_ + 1], p2 = sP[k_ + 2], p3 = sP[k_ + 3];            \
                const v2f q0 = sQ[k_ + 0], q1 = sQ[k_ + 1], q2 = sQ[k_ + 2], q3 = sQ[k_ + 3];            \
                PHD_SEED_PAIR(M, p3, q3); PHD_SEED_PAIR(M, p2, q2); PHD_SEED_PAIR(M, p1, q1); PHD_SEED_PAIR(M, p0, q0); } while (0)
            const int ntop = (nseeds + 7) & ~7;
            for (int r = ntop - 8; r >= 32; r -= 8) PHD_SEED_OCTET(mhi, r);
            for (int r = (ntop < 32 ? ntop : 32) - 8; r >= 0; r -= 8) PHD_SEED_OCTET(mlo, r);
#undef PHD_SEED_OCTET
#undef PHD_SEED_PAIR
            u64 m = ((u64)mhi << 32) | mlo;
            m &= (nseeds == 64) ? ~0ull : ((1ull << nseeds) - 1ull);   // (inf - inf in the padding has no defined sign)
            if (q == 0) RSTAMP(2);
            bool merged = false;
#ifdef PHD_ASSIGN_STATS   // (tools/phase_profile.py prints them; global atomics: they distort the stamps)
            if (STAMPS && ev) { atomicAdd((unsigned long long*)&st[25], (unsigned long long)__popcll(m)); atomicAdd((unsigned long long*)&st[28], 1ull); }
#endif
            if (ev) {
                while (m) {
                    const int r = __builtin_ctzll(m);
                    m &= m - 1;
#ifdef PHD_ASSIGN_STATS
                    if (STAMPS) atomicAdd((unsigned long long*)&st[26], 1ull);
#endif
                    const float f2x = sPf[4 * (r >> 1) + 2 + (r & 1)], f2y = sQf[r];   // the seed's mean is -(−2 m)/2, exactly
                    const v4f g = sG[r];
                    if (is_close<HELLINGER>(-0.5f * f2x, -0.5f * f2y, g.x, g.y, g.z, ea.x, ea.y, fb.x, fb.y, fb.z, T)) {
                        asg[4 * i + 3] = __float_as_int(g.w);
                        merged = true;
#ifdef PHD_ASSIGN_STATS
                        if (STAMPS) atomicAdd((unsigned long long*)&st[27], 1ull);
#endif
                        break;
                    }
                }
            }
            if (ev && !merged) { kept++; keepbits |= 1u << q; }
        }
#ifdef PHD_DUP_ASSIGN
        }
#endif
        RSTAMP(3);
        // ordered compaction of the list: exclusive scan of `kept` over the workgroup (wave scan + wave totals)
        {
            const int incl = (int)wave_incl_scan((u32)kept);
            if (lane == 63) L.ctr[CTR_TMP + wave] = incl;
            __syncthreads();
            RSTAMP(4);
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < PHD_NW; ++w) {
                const int c = L.ctr[CTR_TMP + w];
                if (w < wave) woff += c;
                total += c;
            }
            int o = woff + incl - kept;
            if (total <= tail_cap && total > 64) {                  // (uniform) the one-shot finish follows: stage ITS arrays instead of a window
                const int rw_ = tail_cap >> 6;
                LDS_T(u64)* const trows = (LDS_T(u64)*)L.rwin;      // (the window and the seed records are dead: every wave is past the barrier above)
                LDS_T(v4f)* const tA_ = (LDS_T(v4f)*)(trows + rw_ * tail_cap);
                LDS_T(v4f)* const tB_ = tA_ + tail_cap;
                for (int q = 0; q < per; ++q)
                    if ((keepbits >> q) & 1u) {
                        const int i = cur[64 + tid * per + q];
                        tA_[o] = gA[i]; tB_[o] = gB[i];
                        nxt[o++] = (u16)i;
                    }
                for (int t = tid; t < total * rw_; t += PHD_T) trows[t] = 0ull;
            } else {
            for (int q = 0; q < per; ++q)
                if ((keepbits >> q) & 1u) {
                    const int i = cur[64 + tid * per + q];
                    if (o < 64) { wpos[o] = i; wA[o] = gA[i]; wB[o] = gB[i]; }   // the next round's window
                    nxt[o++] = (u16)i;
                }
            if (tid >= total && tid < 64) ((LDS_T(float)*)&wA[tid])[2] = INFINITY; // padding: never a candidate
            }
            n_u = total;
            __syncthreads();
            lds_u16 t2 = cur; cur = nxt; nxt = t2;
        }
        if (STAMPS && tid == 0) {
            const u64 tq3 = __builtin_amdgcn_s_memrealtime();
            st[12] += tq1 - tq0; st[13] += tq2 - tq1; st[14] += tq3 - tq2; st[15] += 1;
            // finer (thread 0's wave): - | matrix filter | matrix exact + barrier | (resolve: st[13]) | seed records + first pass
            // of the filter over the seeds | exact decisions (+ later passes) | wait for the slowest wave | compaction
            st[16] += tf[0] - tq0; st[17] += tf[1] - tf[0]; st[18] += tq1 - tf[1];
            st[19] += tf[2] - tq2; st[20] += tf[3] - tf[2]; st[21] += tf[4] - tf[3]; st[22] += tq3 - tf[4];
        }
#undef RSTAMP
    }

    STAMP(7);
    // ---- moment matching by exact, order-free sums (moment_sums above) ----
    moment_sums<HELLINGER, STAMPS>(gA, gB, S, kbase, cap, L.acc, L.acc_clusters, L.ctr, T, out_slab, tid, st);
    __syncthreads();
    STAMP(10);
}

} // namespace phd
