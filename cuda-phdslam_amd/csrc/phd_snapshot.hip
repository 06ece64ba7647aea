// phd_snapshot.hip — the pipelined state snapshot of the drop-in loop (include/phdslam.h: phd_snapshot_capture / _send / _wait).
// recoverSlamState (src/main.cpp:318-361) leaves its results — weighted-mean pose, arg-max index (phd_state_kernel), the arg-max
// particle's map (phd_unpack_one_kernel) — in small device buffers; this kernel packs them, every particle's pose and log-weight
// into ONE contiguous staging block, so that the filter's stream can go on with the resample and the next step while a second
// stream downloads the block with one copy.  Plain copies: HBM-bound, ~150 KB at 4096 particles.
#include <hip/hip_runtime.h>

#include "phd_device.h"

namespace phd {

// staging block (32-bit words): [0..5] expected pose | [6] arg-max particle | [7] map size | [8] particle count | [9..15] - |
// [16..23] the step report (copied after the resample launch, phd_snapshot_send) | map: cap x 7 | poses: n_max x 6 | log-weights: n_max |
// resample indices: n_max (7-line log only)
__global__ __launch_bounds__(256) void phd_snapshot_pack_kernel(const float* __restrict__ state_pose, const int* __restrict__ argmax,
                                                                const int* __restrict__ n_map, const float* __restrict__ map_aos,
                                                                const float* __restrict__ poses, const float* __restrict__ logw,
                                                                int n, int n_max, int cap, float* __restrict__ out)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (tid < 6) out[tid] = state_pose[tid];
    int nm = n_map[0];
    if (nm > cap) nm = cap;
    if (tid == 6) ((int*)out)[6] = argmax[0];
    if (tid == 7) ((int*)out)[7] = n_map[0];
    if (tid == 8) ((int*)out)[8] = n;
    float* const o_map = out + PHD_SNAP_HEADER_WORDS;
    float* const o_pose = o_map + (size_t)7 * cap;
    float* const o_logw = o_pose + (size_t)6 * n_max;
    for (int i = tid; i < 7 * nm; i += nt) o_map[i] = map_aos[i];
    for (int i = tid; i < 6 * n; i += nt) o_pose[i] = poses[i];
    for (int i = tid; i < n; i += nt) o_logw[i] = logw[i];
}

hipError_t launch_snapshot_pack(const float* state_pose, const int* argmax, const int* n_map, const phd_gaussian2d* map_aos,
                                const phd_pose* poses, const float* logw, int n, int n_max, int cap, float* out, hipStream_t st)
{
    const int blocks = (6 * n + 7 * cap + 255) / 256 < 64 ? (6 * n + 7 * cap + 255) / 256 : 64;
    hipLaunchKernelGGL(phd_snapshot_pack_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, state_pose, argmax, n_map,
                       (const float*)map_aos, (const float*)poses, logw, n, n_max, cap, out);
    return hipGetLastError();
}

} // namespace phd
