// phd_defs.h — compile-time shape of the workgroup, scalar/vector typedefs and the LDS pointer qualifier of the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "phd_device.h"
#include "phd_detexp.h"

namespace phd {


#ifndef PHD_NW
#define PHD_NW 8            // waves per workgroup (512 threads: two waves per SIMD hide LDS/ALU latency
                            // when a CU holds a single particle; throughput-neutral at 4096 particles)
#endif
#define PHD_T (64 * PHD_NW)
static_assert(PHD_NW == 8, "the kernels are written, tested and tuned for 8 waves per workgroup (4 measured slower and is not maintained)");
#ifndef PHD_MIN_WAVES
#define PHD_MIN_WAVES 4      // launch bound: waves per SIMD the register allocation must allow
#endif
#define PHD_COLS (64 / PHD_NW) // window columns (= candidate seeds) owned by one wave
#define PHD_SMALL_S 256        // survivor counts up to this take the single-shot merge (merge_small)
#define NEAR_U_BASE 0x40000000
// phase stamps of the diagnostic instantiation (100 MHz s_memrealtime), thread 0 of each workgroup
#define STAMP(k) do { if (STAMPS && tid == 0) st[k] = __builtin_amdgcn_s_memrealtime(); } while (0)

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define LDS_T(T) __attribute__((address_space(3))) T

} // namespace phd
