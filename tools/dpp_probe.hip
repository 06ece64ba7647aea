// probe: which DPP / permlane-swap encodings reproduce __shfl_xor(v, off) and __shfl_up(v, 1) on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

template <int CTRL, int RM, int BM>
__device__ u32 dpp(u32 old, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, RM, BM, false); }

__global__ void probe(u32* out)
{
    const u32 lane = threadIdx.x;
    const u32 v = lane * 3 + 1000;
    int k = 0;
    out[(k++) * 64 + lane] = dpp<0xB1, 0xF, 0xF>(v, v);                       // xor 1
    out[(k++) * 64 + lane] = dpp<0x4E, 0xF, 0xF>(v, v);                       // xor 2
    { u32 r = dpp<0x12C, 0xF, 0x5>(v, v); r = dpp<0x124, 0xF, 0xA>(r, v); out[(k++) * 64 + lane] = r; }   // xor 4 (A)
    { u32 r = dpp<0x124, 0xF, 0x5>(v, v); r = dpp<0x12C, 0xF, 0xA>(r, v); out[(k++) * 64 + lane] = r; }   // xor 4 (B)
    out[(k++) * 64 + lane] = dpp<0x128, 0xF, 0xF>(v, v);                      // xor 8
    { u32x2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false); out[(k++) * 64 + lane] = (lane & 16) ? r.x : r.y; }  // xor 16
    { u32x2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false); out[(k++) * 64 + lane] = (lane & 32) ? r.x : r.y; }  // xor 32
    out[(k++) * 64 + lane] = dpp<0x138, 0xF, 0xF>(v, v);                      // wave_shr:1 (shfl_up 1)
    out[(k++) * 64 + lane] = dpp<0x130, 0xF, 0xF>(v, v);                      // wave_shl:1
    out[(k++) * 64 + lane] = dpp<0x141, 0xF, 0xF>(v, v);                      // row_half_mirror
    out[(k++) * 64 + lane] = dpp<0x140, 0xF, 0xF>(v, v);                      // row_mirror
    // references
    for (int off = 1; off < 64; off <<= 1) out[(k++) * 64 + lane] = __shfl_xor(v, off);
    out[(k++) * 64 + lane] = __shfl_up(v, 1);
}

int main()
{
    u32* d; hipMalloc(&d, 64 * 32 * 4);
    hipMemset(d, 0, 64 * 32 * 4);
    probe<<<1, 64>>>(d);
    u32 h[32][64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[11] = {"xor1", "xor2", "xor4A", "xor4B", "xor8", "xor16", "xor32", "wave_shr1", "wave_shl1", "half_mirror", "mirror"};
    const int ref[11] = {11, 12, 13, 13, 14, 15, 16, 17, 17, -1, -1};
    for (int k = 0; k < 11; ++k) {
        int bad = 0;
        if (ref[k] >= 0) for (int l = (k >= 7 ? 1 : 0); l < 64; ++l) bad += h[k][l] != h[ref[k]][l];
        printf("%-12s mismatches %d : ", names[k], bad);
        for (int l = 0; l < 20; ++l) printf("%d ", (int)(h[k][l] - 1000) / 3);
        printf("... %d %d\n", (int)(h[k][32] - 1000) / 3, (int)(h[k][63] - 1000) / 3);
    }
    return 0;
}
