// phd_eap.hip — the expected-a-posteriori (EAP) map estimate on the device.
//
// Reference behaviour (host code there): computeExpectedMap (src/main.cpp:290-316) concatenates
// every particle's map with the feature weights multiplied by exp(particle log-weight) and hands
// the concatenation (sum of all map sizes: 10^4 .. 10^7 Gaussians) to reduceGaussianMixture
// (src/gm_reduce.cpp:57-134): sort by weight, repeatedly take the heaviest unmerged Gaussian as
// a seed, absorb every later Gaussian whose Cholesky-form Mahalanobis distance to the seed
// (:30-37) is below min_distance, moment-match the cluster (:103-129).  O(T * K) distance
// evaluations on one CPU thread in the reference.
//
// Here:  concat kernel (one workgroup per particle, weights scaled by det_exp) ->
//        rocPRIM radix sort (weight desc, stable = index tie-break) -> gather into sorted planes ->
//        ROUNDS over the still-unmerged list (kept compact, in sorted order):
//           window kernel : the first 64 unmerged candidates, 64x64 distances in one wave,
//                           sequential seed resolution on wave-uniform masks
//           assign kernel : every other unmerged Gaussian joins the FIRST seed it is close to
//                           (cheap one-axis bound, then the reference's exact expression)
//           rocPRIM select: the unassigned ones, order preserved, become the next round's list
//        -> stable radix sort by cluster id -> one wave per cluster accumulates its members in
//        the reference's order (seed, then weight-descending), so sums round identically.
// Every Gaussian is tested against each seed that precedes its cluster's seed, exactly like the
// sequential algorithm, so the clustering is the reference's, not an approximation of it.
//
// rocPRIM (header-only, ships with ROCm) is used for the device-wide sort / compaction; it is
// state extraction, not the per-step hot path, which stays free of library calls.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "phd_detexp.h"
#include "phd_device.h"

#pragma clang fp contract(off)

namespace phd {

struct GmPlanes {
    const float* w;
    const float* mx;
    const float* my;
    const float* c00;
    const float* c10;
    const float* c01;
    const float* c11;
};

struct GmSeeds {
    int n;           // seeds of the current round
    unsigned base;   // cluster id of this round's first seed
    unsigned next;   // cluster id the next round starts at (= clusters so far)
    int pad;
    float mx[64], my[64], c00[64], c10[64], c11[64];
};

// mahalanobisDistance(GaussianX, GaussianX), src/gm_reduce.cpp:30-37, with Eigen's LLT of the
// 2x2 mean covariance and the triangular solve written out (oracle: o_chol_dist)
__device__ __forceinline__ float chol_dist(float ax, float ay, float a00, float a10, float a11, float bx, float by,
                                           float b00, float b10, float b11)
{
    const float d0 = ax - bx, d1 = ay - by;
    const float s00 = 0.5f * (a00 + b00);
    const float s10 = 0.5f * (a10 + b10);
    const float s11 = 0.5f * (a11 + b11);
    const float l00 = sqrtf(s00);
    const float l10 = s10 / l00;
    const float l11 = sqrtf(s11 - l10 * l10);
    const float x0 = d0 / l00;
    const float x1 = (d1 - l10 * x0) / l11;
    return x0 * x0 + x1 * x1;
}

__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// ---------------------------------------------------------------------------------------------
// concat: out[k][offset(p) + i] = slab(parent[p])[k][i], weight plane scaled by exp(logw[p])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eap_concat_kernel(const float* __restrict__ maps, const int* __restrict__ counts,
                                                         const int* __restrict__ parent, const float* __restrict__ logw,
                                                         const int* __restrict__ offsets, int cap, float* __restrict__ out,
                                                         size_t T)
{
    const int p = blockIdx.x;
    const int row = parent[p];
    const int cnt = counts[row];
    const size_t off = (size_t)offsets[p];
    const float f = (float)det_exp(logw[p]);                       // map[i].weight *= exp(weights[n]) (src/main.cpp:303)
    const float* slab = maps + (size_t)row * 6 * cap;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        out[off + i] = slab[i] * f;
#pragma unroll
        for (int k = 1; k < 6; ++k) out[(size_t)k * T + off + i] = slab[(size_t)k * cap + i];
    }
}

__global__ void gm_iota_kernel(unsigned* a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (unsigned)i;
}

// sorted planes <- input planes gathered through the sort permutation; list <- 0..T-1
__global__ void gm_gather_kernel(GmPlanes in, const unsigned* __restrict__ perm, float* __restrict__ sp, size_t T, int sym,
                                 unsigned* __restrict__ list)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const unsigned s = perm[i];
    sp[i] = in.w[s];
    sp[T + i] = in.mx[s];
    sp[2 * T + i] = in.my[s];
    sp[3 * T + i] = in.c00[s];
    sp[4 * T + i] = in.c10[s];
    sp[5 * T + i] = in.c11[s];
    if (!sym) sp[6 * T + i] = in.c01[s];
    list[i] = (unsigned)i;
}

// ---------------------------------------------------------------------------------------------
// one round, part 1: the first nw <= 64 unmerged Gaussians.  Lane l holds candidate l.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gm_window_kernel(const unsigned* __restrict__ list, int nw, GmPlanes sp, float min_d,
                                                       GmSeeds* S, unsigned* __restrict__ cluster)
{
    const int l = threadIdx.x;
    const bool ok = l < nw;
    const unsigned pos = ok ? list[l] : 0u;
    const float mx = sp.mx[pos], my = sp.my[pos], c00 = sp.c00[pos], c10 = sp.c10[pos], c11 = sp.c11[pos];
    unsigned long long close_to = 0;                                // earlier candidates this one is close to
    for (int j = 0; j < nw; ++j) {
        const float d = chol_dist(lane_f(mx, j), lane_f(my, j), lane_f(c00, j), lane_f(c10, j), lane_f(c11, j), mx, my, c00,
                                  c10, c11);
        if (ok && j < l && d < min_d) close_to |= 1ull << j;
    }
    // candidate j is a seed iff no earlier SEED absorbed it (src/gm_reduce.cpp:79-100 in sequence)
    unsigned long long seeds = 0;
    for (int j = 0; j < nw; ++j) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)close_to, j);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(close_to >> 32), j);
        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
        if ((m & seeds) == 0) seeds |= 1ull << j;
    }
    const unsigned base = S->next;
    if (ok) {
        const bool is_seed = (seeds >> l) & 1ull;
        const int j = is_seed ? l : __ffsll((long long)(close_to & seeds)) - 1;
        const int rank = __popcll(seeds & ((1ull << j) - 1ull));
        cluster[pos] = base + (unsigned)rank;
        if (is_seed) {
            S->mx[rank] = mx; S->my[rank] = my; S->c00[rank] = c00; S->c10[rank] = c10; S->c11[rank] = c11;
        }
    }
    __syncthreads();
    if (l == 0) {
        const int ns = __popcll(seeds);
        S->n = ns;
        S->base = base;
        S->next = base + (unsigned)ns;
    }
}

// one round, part 2: list entries behind the window against this round's seeds, in seed order
__global__ __launch_bounds__(256) void gm_assign_kernel(const unsigned* __restrict__ list, int n, GmPlanes sp, float min_d,
                                                        const GmSeeds* __restrict__ S, unsigned* __restrict__ cluster,
                                                        unsigned char* __restrict__ keep)
{
    __shared__ float s_mx[64], s_my[64], s_c00[64], s_c10[64], s_c11[64];
    const int ns = S->n;
    const unsigned base = S->base;
    if ((int)threadIdx.x < ns) {
        const int t = threadIdx.x;
        s_mx[t] = S->mx[t]; s_my[t] = S->my[t]; s_c00[t] = S->c00[t]; s_c10[t] = S->c10[t]; s_c11[t] = S->c11[t];
    }
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const unsigned pos = list[e];
    const float bx = sp.mx[pos], by = sp.my[pos], b00 = sp.c00[pos], b10 = sp.c10[pos], b11 = sp.c11[pos];
    const float bound = 1.001f * min_d;
    int hit = -1;
    for (int j = 0; j < ns; ++j) {
        // d = x0^2 + x1^2 >= x0^2 = d0^2 / s00 (to a few ulp): a pair failing this cannot pass the
        // exact test; anything undecided (including NaN operands) takes the reference's expression
        const float d0 = s_mx[j] - bx;
        if (d0 * d0 > bound * (0.5f * (s_c00[j] + b00))) continue;
        const float d = chol_dist(s_mx[j], s_my[j], s_c00[j], s_c10[j], s_c11[j], bx, by, b00, b10, b11);
        if (d < min_d) { hit = j; break; }
    }
    if (hit >= 0) cluster[pos] = base + (unsigned)hit;
    keep[e] = hit < 0;
}

// segment starts of the cluster-sorted order
__global__ void gm_segments_kernel(const unsigned* __restrict__ ckey, size_t T, unsigned K, unsigned* __restrict__ seg)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const unsigned k = ckey[i];
    if (i == 0 || ckey[i - 1] != k) seg[k] = (unsigned)i;
    if (i == 0) seg[K] = (unsigned)T;
}

// moment matching of one cluster per wave, members in (seed, weight-descending) order —
// src/gm_reduce.cpp:103-129; the running sums live in wave-uniform registers
__global__ __launch_bounds__(256) void gm_accumulate_kernel(const unsigned* __restrict__ cpos, const unsigned* __restrict__ seg,
                                                            unsigned K, GmPlanes sp, phd_gaussian2d* __restrict__ out)
{
    const unsigned k = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    const unsigned b = seg[k], e = seg[k + 1];
    const unsigned ps = cpos[b];
    const float ws = sp.w[ps], xs = sp.mx[ps], ys = sp.my[ps];
    float W = ws, m0 = xs * ws, m1 = ys * ws;                                     // :103-104
    for (unsigned c = b + 1; c < e; c += 64) {
        const unsigned i = c + lane;
        const bool ok = i < e;
        const unsigned p = ok ? cpos[i] : ps;
        const float wv = sp.w[p], xv = sp.mx[p], yv = sp.my[p];
        const int cnt = (int)min(64u, e - c);
        for (int l = 0; l < cnt; ++l) {                                           // :105-108
            const float wl = lane_f(wv, l);
            m0 += wl * lane_f(xv, l);
            m1 += wl * lane_f(yv, l);
            W += wl;
        }
    }
    m0 /= W; m1 /= W;                                                             // :109
    float d0 = m0 - xs, d1 = m1 - ys;                                             // :110
    float c00 = ws * (sp.c00[ps] + d0 * d0);                                      // :111-112
    float c10 = ws * (sp.c10[ps] + d1 * d0);
    float c01 = ws * (sp.c01[ps] + d0 * d1);
    float c11 = ws * (sp.c11[ps] + d1 * d1);
    for (unsigned c = b + 1; c < e; c += 64) {
        const unsigned i = c + lane;
        const bool ok = i < e;
        const unsigned p = ok ? cpos[i] : ps;
        const float wv = sp.w[p], xv = sp.mx[p], yv = sp.my[p];
        const float v00 = sp.c00[p], v10 = sp.c10[p], v01 = sp.c01[p], v11 = sp.c11[p];
        const int cnt = (int)min(64u, e - c);
        for (int l = 0; l < cnt; ++l) {                                           // :114-118
            const float wl = lane_f(wv, l);
            d0 = m0 - lane_f(xv, l);
            d1 = m1 - lane_f(yv, l);
            c00 += wl * (lane_f(v00, l) + d0 * d0);
            c10 += wl * (lane_f(v10, l) + d1 * d0);
            c01 += wl * (lane_f(v01, l) + d0 * d1);
            c11 += wl * (lane_f(v11, l) + d1 * d1);
        }
    }
    if (lane == 0) {
        phd_gaussian2d g;
        g.weight = W;
        g.mean[0] = m0; g.mean[1] = m1;
        g.cov[0] = c00 / W; g.cov[1] = c10 / W; g.cov[2] = c01 / W; g.cov[3] = c11 / W;   // :119-129
        out[k] = g;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define EAPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) return e_;                                                                               \
    } while (0)

struct GmWorkspace {
    size_t cap = 0;
    float* concat = nullptr;   // [6][concat_cap]
    size_t concat_cap = 0;
    float* key[2] = {nullptr, nullptr};
    unsigned* val[2] = {nullptr, nullptr};
    float* sp = nullptr;       // [7][cap] planes in weight-descending order
    unsigned* list[2] = {nullptr, nullptr};
    unsigned char* keep = nullptr;
    unsigned* cluster = nullptr;
    unsigned* seg = nullptr;   // [cap + 1]
    GmSeeds* seeds = nullptr;
    unsigned* d_count = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    phd_gaussian2d* d_out = nullptr;
    size_t out_cap = 0;
    int* d_offsets = nullptr;
    size_t offsets_cap = 0;
};

GmWorkspace* gm_workspace_create() { return new GmWorkspace(); }

static void gm_free_all(GmWorkspace* w)
{
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(w->key[i]); (void)hipFree(w->val[i]); (void)hipFree(w->list[i]);
        w->key[i] = nullptr; w->val[i] = nullptr; w->list[i] = nullptr;
    }
    (void)hipFree(w->sp); (void)hipFree(w->keep); (void)hipFree(w->cluster); (void)hipFree(w->seg); (void)hipFree(w->temp);
    w->sp = nullptr; w->keep = nullptr; w->cluster = nullptr; w->seg = nullptr; w->temp = nullptr;
    w->cap = 0; w->temp_bytes = 0;
}

void gm_workspace_destroy(GmWorkspace* w)
{
    if (!w) return;
    gm_free_all(w);
    (void)hipFree(w->concat); (void)hipFree(w->seeds); (void)hipFree(w->d_count); (void)hipFree(w->d_out);
    (void)hipFree(w->d_offsets);
    delete w;
}

static hipError_t gm_grow(GmWorkspace* w, size_t T, hipStream_t st)
{
    if (!w->seeds) {
        EAPCHK(hipMalloc(&w->seeds, sizeof(GmSeeds)));
        EAPCHK(hipMalloc(&w->d_count, sizeof(unsigned)));
    }
    if (T <= w->cap) return hipSuccess;
    EAPCHK(hipStreamSynchronize(st));
    gm_free_all(w);
    const size_t cap = std::max<size_t>(T + T / 4, 1024);
    for (int i = 0; i < 2; ++i) {
        EAPCHK(hipMalloc(&w->key[i], cap * sizeof(float)));
        EAPCHK(hipMalloc(&w->val[i], cap * sizeof(unsigned)));
        EAPCHK(hipMalloc(&w->list[i], cap * sizeof(unsigned)));
    }
    EAPCHK(hipMalloc(&w->sp, 7 * cap * sizeof(float)));
    EAPCHK(hipMalloc(&w->keep, cap));
    EAPCHK(hipMalloc(&w->cluster, cap * sizeof(unsigned)));
    EAPCHK(hipMalloc(&w->seg, (cap + 1) * sizeof(unsigned)));
    size_t b1 = 0, b2 = 0, b3 = 0;
    EAPCHK(rocprim::radix_sort_pairs_desc(nullptr, b1, w->key[0], w->key[1], w->val[0], w->val[1], cap, 0, 32, st));
    EAPCHK(rocprim::radix_sort_pairs(nullptr, b2, w->cluster, w->val[0], w->val[0], w->val[1], cap, 0, 32, st));
    EAPCHK(rocprim::select(nullptr, b3, w->list[0], w->keep, w->list[1], w->d_count, cap, st));
    w->temp_bytes = std::max(b1, std::max(b2, b3)) + 256;
    EAPCHK(hipMalloc(&w->temp, w->temp_bytes));
    w->cap = cap;
    return hipSuccess;
}

float* gm_concat_buffer(GmWorkspace* w, size_t T, hipStream_t st)
{
    if (T > w->concat_cap) {
        if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
        (void)hipFree(w->concat);
        w->concat = nullptr;
        w->concat_cap = 0;
        const size_t cap = std::max<size_t>(T + T / 4, 1024);
        if (hipMalloc(&w->concat, 6 * cap * sizeof(float)) != hipSuccess) return nullptr;
        w->concat_cap = cap;
    }
    return w->concat;
}

int* gm_offsets_buffer(GmWorkspace* w, size_t n, hipStream_t st)
{
    if (n > w->offsets_cap) {
        if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
        (void)hipFree(w->d_offsets);
        w->d_offsets = nullptr;
        w->offsets_cap = 0;
        if (hipMalloc(&w->d_offsets, n * sizeof(int)) != hipSuccess) return nullptr;
        w->offsets_cap = n;
    }
    return w->d_offsets;
}

hipError_t launch_eap_concat(const float* maps, const int* counts, const int* parent, const float* logw, const int* offsets,
                             int cap, int n, float* out, size_t T, hipStream_t st)
{
    if (n <= 0 || T == 0) return hipSuccess;
    eap_concat_kernel<<<n, 256, 0, st>>>(maps, counts, parent, logw, offsets, cap, out, T);
    return hipGetLastError();
}

// reduceGaussianMixture over T Gaussians given as SoA planes in device memory (in[5] == in[4] for
// symmetric covariances).  Result: ws->d_out[0..K) on the device, K and the round count returned.
hipError_t gm_reduce_device(GmWorkspace* w, const float* const in[7], size_t T, float min_distance, hipStream_t st,
                            int* K_out, int* rounds_out, const phd_gaussian2d** d_result)
{
    *K_out = 0;
    if (rounds_out) *rounds_out = 0;
    if (d_result) *d_result = nullptr;
    if (T == 0) return hipSuccess;
    if (T > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    EAPCHK(gm_grow(w, T, st));
    const GmPlanes pin = {in[0], in[1], in[2], in[3], in[4], in[5], in[6]};
    const int sym = in[5] == in[4];
    const unsigned nb = (unsigned)((T + 255) / 256);
    // 1. sort by weight, descending, stable (std::sort + compare_gaussians, :75-77)
    EAPCHK(hipMemcpyAsync(w->key[0], in[0], T * sizeof(float), hipMemcpyDeviceToDevice, st));
    gm_iota_kernel<<<nb, 256, 0, st>>>(w->val[0], T);
    size_t tb = w->temp_bytes;
    EAPCHK(rocprim::radix_sort_pairs_desc(w->temp, tb, w->key[0], w->key[1], w->val[0], w->val[1], T, 0, 32, st));
    gm_gather_kernel<<<nb, 256, 0, st>>>(pin, w->val[1], w->sp, T, sym, w->list[0]);
    EAPCHK(hipGetLastError());
    float* s = w->sp;
    const GmPlanes sp = {s, s + T, s + 2 * T, s + 3 * T, s + 4 * T, sym ? s + 4 * T : s + 6 * T, s + 5 * T};
    // 2. rounds
    EAPCHK(hipMemsetAsync(w->seeds, 0, sizeof(GmSeeds), st));
    unsigned* la = w->list[0];
    unsigned* lb = w->list[1];
    size_t nR = T;
    int rounds = 0;
    while (nR > 0) {
        const int nw = (int)std::min<size_t>(nR, 64);
        gm_window_kernel<<<1, 64, 0, st>>>(la, nw, sp, min_distance, w->seeds, w->cluster);
        ++rounds;
        const size_t rest = nR - nw;
        if (rest == 0) break;
        gm_assign_kernel<<<(unsigned)((rest + 255) / 256), 256, 0, st>>>(la + nw, (int)rest, sp, min_distance, w->seeds,
                                                                         w->cluster, w->keep);
        EAPCHK(hipGetLastError());
        tb = w->temp_bytes;
        EAPCHK(rocprim::select(w->temp, tb, la + nw, w->keep, lb, w->d_count, rest, st));
        unsigned cnt = 0;
        EAPCHK(hipMemcpyAsync(&cnt, w->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        EAPCHK(hipStreamSynchronize(st));
        nR = cnt;
        std::swap(la, lb);
    }
    unsigned K = 0;
    EAPCHK(hipMemcpyAsync(&K, &w->seeds->next, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    EAPCHK(hipStreamSynchronize(st));
    if (K == 0 || K > T) return hipErrorUnknown;
    // 3. group by cluster (stable: seed first, then weight-descending) and moment-match
    int bits = 1;
    while ((1ull << bits) < K) ++bits;
    gm_iota_kernel<<<nb, 256, 0, st>>>(w->val[0], T);
    unsigned* ckey = reinterpret_cast<unsigned*>(w->key[0]);
    tb = w->temp_bytes;
    EAPCHK(rocprim::radix_sort_pairs(w->temp, tb, w->cluster, ckey, w->val[0], w->val[1], T, 0, bits, st));
    gm_segments_kernel<<<nb, 256, 0, st>>>(ckey, T, K, w->seg);
    if (K > w->out_cap) {
        EAPCHK(hipStreamSynchronize(st));
        (void)hipFree(w->d_out);
        w->d_out = nullptr;
        w->out_cap = 0;
        EAPCHK(hipMalloc(&w->d_out, (size_t)(K + K / 2 + 64) * sizeof(phd_gaussian2d)));
        w->out_cap = K + K / 2 + 64;
    }
    gm_accumulate_kernel<<<(K + 3) / 4, 256, 0, st>>>(w->val[1], w->seg, K, sp, w->d_out);
    EAPCHK(hipGetLastError());
    *K_out = (int)K;
    if (rounds_out) *rounds_out = rounds;
    if (d_result) *d_result = w->d_out;
    return hipSuccess;
}

} // namespace phd
