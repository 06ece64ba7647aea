// phd_snapshot.hip — the pipelined state snapshot of the drop-in loop (include/phdslam.h: phd_snapshot_capture / _send / _wait).
// ONE launch does what phd_state_snapshot does with two kernels and six downloads: recoverSlamState (src/main.cpp:318-361) —
// weighted-mean pose, arg-max particle (ties to the lowest index), that particle's map — and every particle's pose and
// log-weight, packed into ONE contiguous staging block, so that the filter's stream can go on with the resample and the next
// step while a second stream downloads the block with one copy.  Plain copies beside one reduction: ~150 KB at 4096 particles.
//
// Workgroup 0 is phd_state_kernel + phd_unpack_one_kernel (phd_kernels.hip) restated operation for operation — the same strided
// partial sums, the same butterfly, the same order over the waves, the same libm expf — so the expected pose has the same bits
// (tests/test_gpu_driver.py: the pipelined loop's files equal the synchronous loop's byte for byte; ..._slots_equal_the_blocking_snapshot).
#include <hip/hip_runtime.h>
#include <float.h>

#include "phd_device.h"
#include "phd_lane.h"

namespace phd {

#define PHD_SNAP_T 1024           // = PHD_WT of phd_state_kernel: the reduction tree is a function of the block size

__device__ __forceinline__ float snap_block_sum(float v, float* sc, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + xor_lane(v, off);
    __syncthreads();
    if ((tid & 63) == 0) sc[tid >> 6] = v;
    __syncthreads();
    float r = sc[0];
    for (int w = 1; w < PHD_SNAP_T / 64; ++w) r = r + sc[w];
    return r;
}

// staging block (32-bit words): [0..5] expected pose | [6] arg-max particle | [7] map size | [8] particle count | [9..15] - |
// [16..23] the step report (copied after the resample launch, phd_snapshot_send) | map: cap x 7 | CPHD: the arg-max particle's cardinality row,
// cn_len | poses: n_max x 6 | log-weights: n_max | resample indices: n_max (7-line log only)
__global__ __launch_bounds__(PHD_SNAP_T) void phd_snapshot_kernel(const phd_pose* __restrict__ poses, const float* __restrict__ logw, int n,
                                                                  int n_max, const float* __restrict__ slabs, const int* __restrict__ parent,
                                                                  const int* __restrict__ counts, int cap, const float* __restrict__ cn, int cn_len,
                                                                  float* __restrict__ out)
{
    const int tid = threadIdx.x;
    float* const o_map = out + PHD_SNAP_HEADER_WORDS;
    float* const o_cn = o_map + (size_t)7 * cap;
    float* const o_pose = o_cn + cn_len;
    float* const o_logw = o_pose + (size_t)6 * n_max;
    if (blockIdx.x > 0) {
        // ---- the copies: every particle's pose and log-weight
        const int t = (blockIdx.x - 1) * PHD_SNAP_T + tid, nt = (gridDim.x - 1) * PHD_SNAP_T;
        const float* pf = (const float*)poses;
        for (int i = t; i < 6 * n; i += nt) o_pose[i] = pf[i];
        for (int i = t; i < n; i += nt) o_logw[i] = logw[i];
        return;
    }
    // ---- workgroup 0: phd_state_kernel, then phd_unpack_one_kernel for its arg-max
    __shared__ float sc[PHD_SNAP_T / 64];
    __shared__ float s_best[PHD_SNAP_T / 64];
    __shared__ int s_besti[PHD_SNAP_T / 64];
    __shared__ int s_arg;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    float best = -FLT_MAX;
    int besti = 0x7FFFFFFF;
    for (int i = tid; i < n; i += PHD_SNAP_T) {
        const float lw = logw[i];
        const float w = expf(lw);
        const phd_pose q = poses[i];
        acc[0] += w * q.px; acc[1] += w * q.py; acc[2] += w * q.ptheta;
        acc[3] += w * q.vx; acc[4] += w * q.vy; acc[5] += w * q.vtheta;
        if (lw > best) { best = lw; besti = i; }
    }
    for (int k = 0; k < 6; ++k) {
        const float r = snap_block_sum(acc[k], sc, tid);
        if (tid == 0) out[k] = r;
    }
    // arg-max with ties to the lowest index (strict '>' scan in the reference)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = xor_lane(best, off);
        const int oi = xor_lane(besti, off);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_besti[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < PHD_SNAP_T / 64; ++w)
            if (s_best[w] > best || (s_best[w] == best && s_besti[w] < besti)) { best = s_best[w]; besti = s_besti[w]; }
        if (n == 1) { // src/main.cpp:381-384
            out[0] = poses[0].px; out[1] = poses[0].py; out[2] = poses[0].ptheta;
            out[3] = poses[0].vx; out[4] = poses[0].vy; out[5] = poses[0].vtheta;
        }
        s_arg = (besti == 0x7FFFFFFF) ? -1 : besti;
    }
    __syncthreads();
    const int p = s_arg;
    int cnt = 0;
    if (p >= 0) {
        const int src = parent ? parent[p] : p;
        cnt = counts[src];
        const float* s = slabs + (size_t)src * 6 * cap;
        phd_gaussian2d* const om = (phd_gaussian2d*)o_map;
        for (int i = tid; i < cnt && i < cap; i += PHD_SNAP_T) {
            phd_gaussian2d v;
            v.weight = s[0 * cap + i];
            v.mean[0] = s[1 * cap + i];
            v.mean[1] = s[2 * cap + i];
            v.cov[0] = s[3 * cap + i];
            v.cov[1] = s[4 * cap + i];
            v.cov[2] = s[4 * cap + i];
            v.cov[3] = s[5 * cap + i];
            om[i] = v;
        }
        // CPHD: cn_estimate = the arg-max particle's cardinality row (src/main.cpp:360), indexed like its slab
        for (int i = tid; i < cn_len; i += PHD_SNAP_T) o_cn[i] = cn[(size_t)src * cn_len + i];
    }
    if (tid == 0) { ((int*)out)[6] = p; ((int*)out)[7] = cnt; ((int*)out)[8] = n; }
}

hipError_t launch_snapshot(const phd_pose* poses, const float* logw, int n, int n_max, const float* slabs, const int* parent,
                           const int* counts, int cap, const float* cn, int cn_len, float* out, hipStream_t st)
{
    int copy_blocks = (6 * n + PHD_SNAP_T - 1) / PHD_SNAP_T;
    if (copy_blocks > 32) copy_blocks = 32;
    if (copy_blocks < 1) copy_blocks = 1;
    hipLaunchKernelGGL(phd_snapshot_kernel, dim3(1 + copy_blocks), dim3(PHD_SNAP_T), 0, st, poses, logw, n, n_max, slabs, parent, counts, cap, cn, cn ? cn_len : 0, out);
    return hipGetLastError();
}

} // namespace phd
