// What does one SIMD of gfx950 sustain in vector-ALU ISSUE, and what does the counter formula of tools/pmc_sq.sh
// (valu_issue_fraction = 4 * SQ_ACTIVE_INST_VALU / SIMD-cycles) read on a kernel that is known to saturate it?
//
// Every wave runs a long unrolled stream of INDEPENDENT vector instructions of one class (16 accumulators, so a
// dependent instruction is 16 issues away), at 1, 2, 3, 4, 6 and 8 waves per SIMD (256-thread workgroups = one wave per
// SIMD each, the number resident per CU fixed by the dynamic LDS request and a grid of exactly CUs x n workgroups) and in
// the headline kernel's own launch shape (512 threads, 52 432 B of LDS: three workgroups = 6 waves per SIMD).
// Output: one line per launch - instructions per cycle and SIMD from the waves' own s_memtime stamps and from the
// launch's wall time, the effective shader clock (s_memtime against the 100 MHz s_memrealtime) - and, with
// `--json file`, the launch list in dispatch order for tools/valu_ceiling.py to join with a rocprofv3 --pmc pass.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/valu_issue_probe.hip -o tools/probes/valu_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define R8x(M) R16(M) R16(M) R16(M) R16(M) R16(M) R16(M) R16(M) R16(M)     /* 128 instructions per trip */
#define ACC16 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)

// one entry per class: the text of instruction i of 16 (operands %0..%15 the accumulators, %16 / %17 two inputs)
#define I_FMA(i)    "v_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define I_ADD(i)    "v_add_f32 %" #i ", %16, %" #i "\n"
#define I_PKFMA(i)  "v_pk_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define I_PKADD(i)  "v_pk_add_f32 %" #i ", %16, %" #i "\n"
#define I_ADDU(i)   "v_add_u32 %" #i ", %16, %" #i "\n"
#define I_MOV(i)    "v_mov_b32 %" #i ", %16\n"
#define I_CND(i)    "v_cndmask_b32 %" #i ", %16, %" #i ", vcc\n"
#define I_EXP(i)    "v_exp_f32 %" #i ", %" #i "\n"
#define I_DPP(i)    "v_mov_b32_dpp %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_FMA64(i)  "v_fma_f64 %" #i ", %16, %17, %" #i "\n"
#define I_ADD64(i)  "v_add_f64 %" #i ", %16, %" #i "\n"
#define I_MULLO(i)  "v_mul_lo_u32 %" #i ", %16, %" #i "\n"
#define I_CMP(i)    "v_cmp_lt_f32 vcc, %16, %" #i "\n"
// mixes: instruction i of 16 picks its class from i
#define I_MIX3_0(i) I_CND(i)
#define I_MIX3_1(i) I_MOV(i)
#define I_MIX3_2(i) I_ADDU(i)
#define MIX3 I_CND(0) I_MOV(1) I_ADDU(2) I_CND(3) I_MOV(4) I_ADDU(5) I_CND(6) I_MOV(7) I_ADDU(8) I_CND(9) I_MOV(10) I_ADDU(11) I_CND(12) I_MOV(13) I_ADDU(14) I_CND(15)
#define MIXFI I_FMA(0) I_ADDU(1) I_FMA(2) I_ADDU(3) I_FMA(4) I_ADDU(5) I_FMA(6) I_ADDU(7) I_FMA(8) I_ADDU(9) I_FMA(10) I_ADDU(11) I_FMA(12) I_ADDU(13) I_FMA(14) I_ADDU(15)
// the headline kernel's own proportions (profiles/r04_instruction_mix_cfg3.txt): per 16 vector instructions ~5 fp32
// add/mul/fma, ~4 INT32, ~7 moves / compares / selects / cross-lane, and one scalar instruction per two vector ones
#define SAL "s_add_u32 s20, s20, 1\n"
#define MIXK I_FMA(0) I_ADDU(1) SAL I_MOV(2) I_CND(3) SAL I_FMA(4) I_ADDU(5) SAL I_DPP(6) I_ADD(7) SAL I_MOV(8) I_ADDU(9) SAL I_FMA(10) I_CND(11) SAL I_MOV(12) I_ADDU(13) SAL I_FMA(14) I_CND(15) SAL
#define MIXS I_FMA(0) I_FMA(1) SAL I_FMA(2) I_FMA(3) SAL I_FMA(4) I_FMA(5) SAL I_FMA(6) I_FMA(7) SAL I_FMA(8) I_FMA(9) SAL I_FMA(10) I_FMA(11) SAL I_FMA(12) I_FMA(13) SAL I_FMA(14) I_FMA(15) SAL
#define X8(S) S S S S S S S S

enum { M_FMA, M_ADD, M_PKFMA, M_PKADD, M_MIX3, M_FMA_INT, M_EXP, M_DPP, M_FMA64, M_ADD64, M_MULLO, M_VALU_SALU, M_KERNEL_MIX, M_COUNT };
static const char* mode_name[M_COUNT] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cndmask/v_mov/v_add_u32", "v_fma_f32 : v_add_u32 1:1",
                                         "v_exp_f32", "v_mov_b32_dpp row_shr:1", "v_fma_f64", "v_add_f64", "v_mul_lo_u32", "v_fma_f32 + s_add_u32 2:1",
                                         "headline mix (5 fp32, 4 int, 7 mov/cnd/dpp per 16 + 8 scalar)"};
// vector instructions per 128-entry trip (the scalar ones of the mixes do not count), flops per instruction and lane
static const double mode_flop[M_COUNT] = {2, 1, 4, 2, 0, 1, 1, 0, 2, 1, 0, 2, 0.75};

template <int MODE, int BS>
__global__ __launch_bounds__(BS) void probe(int trips, float seed, unsigned long long* __restrict__ stamps, float* __restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_request[];
    const float b = 1.0f + seed * 1e-9f, c = seed * 1e-9f;
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    float out = 0.f;
    if constexpr (MODE == M_PKFMA || MODE == M_PKADD) {
        f2 a0 = {seed, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, a8 = a0, a9 = a0, a10 = a0, a11 = a0, a12 = a0, a13 = a0, a14 = a0, a15 = a0;
        const f2 b2 = {b, b}, c2 = {c, c};
        asm volatile("s_memrealtime %0\ns_memtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0));
        for (int t = 0; t < trips; ++t) {
            if constexpr (MODE == M_PKFMA) asm volatile(R8x(I_PKFMA) : ACC16 : "v"(b2), "v"(c2));
            else asm volatile(R8x(I_PKADD) : ACC16 : "v"(c2), "v"(c2));
        }
        asm volatile("s_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        out = (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15).x;
    } else if constexpr (MODE == M_FMA64 || MODE == M_ADD64) {
        double a0 = seed, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, a8 = a0, a9 = a0, a10 = a0, a11 = a0, a12 = a0, a13 = a0, a14 = a0, a15 = a0;
        const double bd = b, cd = c;
        asm volatile("s_memrealtime %0\ns_memtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0));
        for (int t = 0; t < trips; ++t) {
            if constexpr (MODE == M_FMA64) asm volatile(R8x(I_FMA64) : ACC16 : "v"(bd), "v"(cd));
            else asm volatile(R8x(I_ADD64) : ACC16 : "v"(cd), "v"(cd));
        }
        asm volatile("s_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        out = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15);
    } else {
        float a0 = seed + threadIdx.x, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, a8 = a0, a9 = a0, a10 = a0, a11 = a0, a12 = a0, a13 = a0, a14 = a0, a15 = a0;
        asm volatile("s_memrealtime %0\ns_memtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0));
        for (int t = 0; t < trips; ++t) {
            if constexpr (MODE == M_FMA) asm volatile(R8x(I_FMA) : ACC16 : "v"(b), "v"(c));
            if constexpr (MODE == M_ADD) asm volatile(R8x(I_ADD) : ACC16 : "v"(c), "v"(c));
            if constexpr (MODE == M_MIX3) asm volatile("v_cmp_lt_f32 vcc, %16, %17\n" X8(MIX3) : ACC16 : "v"(b), "v"(c) : "vcc");
            if constexpr (MODE == M_FMA_INT) asm volatile(X8(MIXFI) : ACC16 : "v"(b), "v"(c));
            if constexpr (MODE == M_EXP) asm volatile(R8x(I_EXP) : ACC16 : "v"(b), "v"(c));
            if constexpr (MODE == M_DPP) asm volatile(R8x(I_DPP) : ACC16 : "v"(b), "v"(c));
            if constexpr (MODE == M_MULLO) asm volatile(R8x(I_MULLO) : ACC16 : "v"(b), "v"(c));
            if constexpr (MODE == M_VALU_SALU) asm volatile(X8(MIXS) : ACC16 : "v"(b), "v"(c) : "s20", "scc");
            if constexpr (MODE == M_KERNEL_MIX) asm volatile("v_cmp_lt_f32 vcc, %16, %17\n" X8(MIXK) : ACC16 : "v"(b), "v"(c) : "s20", "scc", "vcc");
        }
        asm volatile("s_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        out = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
    }
    if ((threadIdx.x & 63) == 0) {      // one record per wave: shader cycles and 100 MHz ticks of its stream
        const size_t w = (size_t)blockIdx.x * (BS / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
    if (out == 12345.678f) sink[0] = out + lds_request[threadIdx.x];
}

struct Launch { int mode, bs, wg_per_cu, lds, grid, trips; double wall_us, wave_cyc, clock_ghz, ipc_wave, ipc_simd_stamps, ipc_simd_wall; };

template <int MODE, int BS> void go(Launch& L, int cus, unsigned long long* d_st, float* d_sink)
{
    hipFuncSetAttribute((const void*)probe<MODE, BS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    probe<MODE, BS><<<L.grid, BS, L.lds, 0>>>(L.trips, 1.0f, d_st, d_sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const size_t waves = (size_t)L.grid * (BS / 64);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (size_t w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
    cyc /= waves; rt /= waves;
    const double instr = 128.0 * L.trips;                        // vector instructions per wave
    const int waves_per_simd = L.wg_per_cu * (BS / 64) / 4;
    L.wall_us = ms * 1e3;
    L.wave_cyc = cyc;
    L.clock_ghz = cyc / (rt * 10.0);                             // s_memrealtime ticks at 100 MHz = 10 ns
    L.ipc_wave = instr / cyc;
    L.ipc_simd_stamps = waves_per_simd * instr / cyc;            // all waves of a SIMD run side by side (one round)
    L.ipc_simd_wall = (double)waves * instr / (ms * 1e-3 * L.clock_ghz * 1e9 * cus * 4);
    hipEventDestroy(a); hipEventDestroy(b);
}

template <int MODE> void go_bs(Launch& L, int cus, unsigned long long* st, float* sink)
{
    if (L.bs == 256) go<MODE, 256>(L, cus, st, sink); else go<MODE, 512>(L, cus, st, sink);
}

int main(int argc, char** argv)
{
    const char* json = nullptr;
    int trips = 4000;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--json") && i + 1 < argc) json = argv[++i];
        if (!strcmp(argv[i], "--trips") && i + 1 < argc) trips = atoi(argv[++i]);
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clockRate %d kHz; %d trips of 128 vector instructions per wave\n", prop.gcnArchName, cus, prop.clockRate, trips);
    unsigned long long* d_st; float* d_sink;
    hipMalloc(&d_st, (size_t)cus * 8 * 8 * 2 * 8);
    hipMalloc(&d_sink, 64);
    // shapes: n workgroups of 256 threads per CU (n waves per SIMD), LDS request 160 KB / n so that no more fit;
    // and the headline's 512 threads x 3 per CU at its own 52 432 B
    struct Shape { int bs, wg_per_cu, lds; };
    std::vector<Shape> shapes;
    for (int n : {1, 2, 3, 4, 6, 8}) shapes.push_back({256, n, (160 * 1024 / n) & ~1023});
    shapes.push_back({512, 3, 52432});
    std::vector<Launch> all;
    {   // one throw-away launch (clocks, code upload) - listed, so the dispatch order of the PMC pass stays aligned
        Launch L{M_FMA, 256, 1, 160 * 1024, cus, 200};
        go_bs<M_FMA>(L, cus, d_st, d_sink);
        all.push_back(L);
    }
    for (int m = 0; m < M_COUNT; ++m)
        for (const Shape& s : shapes) {
            Launch L{m, s.bs, s.wg_per_cu, s.lds, cus * s.wg_per_cu, trips};
            switch (m) {
            case M_FMA: go_bs<M_FMA>(L, cus, d_st, d_sink); break;
            case M_ADD: go_bs<M_ADD>(L, cus, d_st, d_sink); break;
            case M_PKFMA: go_bs<M_PKFMA>(L, cus, d_st, d_sink); break;
            case M_PKADD: go_bs<M_PKADD>(L, cus, d_st, d_sink); break;
            case M_MIX3: go_bs<M_MIX3>(L, cus, d_st, d_sink); break;
            case M_FMA_INT: go_bs<M_FMA_INT>(L, cus, d_st, d_sink); break;
            case M_EXP: go_bs<M_EXP>(L, cus, d_st, d_sink); break;
            case M_DPP: go_bs<M_DPP>(L, cus, d_st, d_sink); break;
            case M_FMA64: go_bs<M_FMA64>(L, cus, d_st, d_sink); break;
            case M_ADD64: go_bs<M_ADD64>(L, cus, d_st, d_sink); break;
            case M_MULLO: go_bs<M_MULLO>(L, cus, d_st, d_sink); break;
            case M_VALU_SALU: go_bs<M_VALU_SALU>(L, cus, d_st, d_sink); break;
            case M_KERNEL_MIX: go_bs<M_KERNEL_MIX>(L, cus, d_st, d_sink); break;
            }
            all.push_back(L);
            const int wps = L.wg_per_cu * (L.bs / 64) / 4;
            printf("%-62s %3d thr x %d/CU = %d waves/SIMD  %9.1f us  clock %.3f GHz  per wave %.4f  per SIMD %.4f (stamps) %.4f (wall) instr/cycle = one per %.2f cycles  %6.1f TFLOP/s\n",
                   mode_name[m], L.bs, L.wg_per_cu, wps, L.wall_us, L.clock_ghz, L.ipc_wave, L.ipc_simd_stamps, L.ipc_simd_wall, 1.0 / L.ipc_simd_stamps,
                   mode_flop[m] * 64 * L.ipc_simd_stamps * L.clock_ghz * 1e9 * cus * 4 * 1e-12);
        }
    if (json) {
        FILE* f = fopen(json, "w");
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"trips\": %d, \"launches\": [\n", prop.gcnArchName, cus, trips);
        for (size_t i = 0; i < all.size(); ++i) {
            const Launch& L = all[i];
            fprintf(f, " {\"seq\": %zu, \"mode\": \"%s\", \"threads\": %d, \"wg_per_cu\": %d, \"waves_per_simd\": %d, \"lds\": %d, \"grid\": %d, \"trips\": %d, \"wall_us\": %.2f, "
                       "\"clock_ghz\": %.4f, \"ipc_wave\": %.5f, \"ipc_simd_stamps\": %.5f, \"ipc_simd_wall\": %.5f}%s\n",
                    i, mode_name[L.mode], L.bs, L.wg_per_cu, L.wg_per_cu * (L.bs / 64) / 4, L.lds, L.grid, L.trips, L.wall_us, L.clock_ghz, L.ipc_wave,
                    L.ipc_simd_stamps, L.ipc_simd_wall, i + 1 < all.size() ? "," : "");
        }
        fprintf(f, "]}\n");
        fclose(f);
    }
    return 0;
}
