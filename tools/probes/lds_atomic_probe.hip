// LDS atomic throughput on gfx950: no-return adds of 64-bit / 32-bit integers and floats to random slots of a 320-entry table
// (what the merge's moment sums do), every CU loaded with three workgroups of 512 threads.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/lds_atomic_probe.hip -o tools/probes/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_T(T) __attribute__((address_space(3))) T
template <int MODE>
__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ idx, int iters, int nslots, double* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LDS_T(double)* d = (LDS_T(double)*)(LDS_T(unsigned char)*)raw;
    LDS_T(unsigned long long)* q = (LDS_T(unsigned long long)*)(LDS_T(unsigned char)*)raw;
    LDS_T(unsigned)* u = (LDS_T(unsigned)*)(LDS_T(unsigned char)*)raw;
    LDS_T(float)* f = (LDS_T(float)*)(LDS_T(unsigned char)*)raw;
    for (int i = threadIdx.x; i < 2 * nslots * 8; i += 512) u[i] = 0;
    __syncthreads();
    unsigned c = idx[threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        const unsigned s = c % (unsigned)nslots;
        if (MODE == 0) __hip_atomic_fetch_add(d + s, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __hip_atomic_fetch_add(q + s, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add(u + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) __hip_atomic_fetch_add(f + s, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) __hip_atomic_fetch_max(u + s, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 5) { // five planes of doubles, as pass A
#pragma unroll
            for (int p = 0; p < 5; ++p) __hip_atomic_fetch_add(d + p * nslots + s, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 6) { // ten planes of 32-bit integers
#pragma unroll
            for (int p = 0; p < 10; ++p) __hip_atomic_fetch_add(u + p * nslots + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 7) { // plain read-modify-write without atomicity (upper bound of what the array can do)
            d[s] = d[s] + 1.0;
        }
        c = c * 1664525u + 1013904223u;
    }
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = d[1];
}

template <int MODE> double run(const unsigned* idx, int iters, int nslots, int grid)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<grid, 512, 2 * nslots * 8 * 4, 0>>>(idx, 8, nslots, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<grid, 512, 2 * nslots * 8 * 4, 0>>>(idx, iters, nslots, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3;
}

int main()
{
    const int nslots = 320, iters = 2000, grid = 768;
    std::vector<unsigned> h(512);
    srand(7);
    for (auto& v : h) v = (unsigned)rand();
    unsigned* idx;
    hipMalloc(&idx, 512 * 4);
    hipMemcpy(idx, h.data(), 512 * 4, hipMemcpyHostToDevice);
    const char* names[] = {"ds_add_f64", "ds_add_u64", "ds_add_u32", "ds_add_f32", "ds_max_u32", "5 x ds_add_f64 (planes)", "10 x ds_add_u32 (planes)", "plain f64 rmw"};
    const double per_it[] = {1, 1, 1, 1, 1, 5, 10, 1};
    double t[8];
    t[0] = run<0>(idx, iters, nslots, grid); t[1] = run<1>(idx, iters, nslots, grid); t[2] = run<2>(idx, iters, nslots, grid);
    t[3] = run<3>(idx, iters, nslots, grid); t[4] = run<4>(idx, iters, nslots, grid); t[5] = run<5>(idx, iters, nslots, grid);
    t[6] = run<6>(idx, iters, nslots, grid); t[7] = run<7>(idx, iters, nslots, grid);
    for (int m = 0; m < 8; ++m) {
        // three workgroups per CU (768 on 256 CUs): lane-operations per CU = 3 * 512 * iters * per_it
        const double lane_ops = 3.0 * 512 * iters * per_it[m];
        printf("%-28s %8.1f us   %.2f lane-operations per cycle per CU (2.4 GHz)\n", names[m], t[m] * 1e6, lane_ops / (t[m] * 2.4e9));
    }
    return 0;
}
