// phd_lds_layout.h — the byte layout of the update+merge kernel's dynamic LDS: plain arithmetic, shared by the device code
// (phd_lds.h carves pointers from it), the host (phd_update_lds_bytes) and tests/test_kernel_resources.py (compiled with g++).
// Needs: PHD_NW, PHD_SMALL_S, u32 (phd_defs.h, or the test's own definitions).
#pragma once

#ifndef PHD_LAYOUT_FN
#define PHD_LAYOUT_FN __host__ __device__ __forceinline__
#endif

namespace phd {

// merge rounds: window copy (pos[64] | gA[64] | gB[64]) + the round's seed records (2 x 72 float4), behind the round lists
#define PHD_RWIN_BYTES (4u * 64u + 2u * 16u * 64u + 2u * 16u * 72u)

// The layout is sized so that THREE workgroups share a CU at 4096 x 256 x 64 (S = 1024, C = 512, MM = 64: 52 432 B + 608 B static,
// three of them 159 KB of the 160): the kernel is latency-bound per workgroup, a third resident one is worth ~20 %
// (profiles/r04_three_workgroups.txt).  What that took, against round 3's 73 KB:
//   * the gain and the updated covariance of a feature are not kept (36 B per feature): the few detection terms that survive the
//     prune rebuild them from the prior feature (detection_posterior(), phd_kernels.hip);
//   * the merge's per-cluster seed records are gone: a survivor's assignment word carries its cluster AND its seed's index;
//   * pass 1's partial sums are stored without the 8-fold padding; the birth geometry waits in registers;
//   * the moment sums' accumulators run over everything that is dead by then (region X and the update's small arrays), and clusters
//     beyond what that holds take a second sweep.
struct LdsOffsets {
    u32 w, mx, my, xx, xy, yy, tr, u;
    u32 alias;      // start of region X: the update's feature terms U the merge's sort arrays / round lists and window
    u32 part, win, z_r, z_b, logZ, zok, zpart;   // the update's small arrays and the rounds' matrix parts, directly behind X
    u32 wide_end;   // [alias, wide_end): what the moment sums' accumulators (and merge_small's rows) may use
    u32 out_idx, red, ctr;
    u32 total;
};

PHD_LAYOUT_FN u32 align16u(u32 x) { return (x + 15u) & ~15u; }

PHD_LAYOUT_FN LdsOffsets lds_offsets(int S, int C, int MM)
{
    LdsOffsets o;
    u32 p = 0;
    const u32 sv = align16u(4u * (u32)S);
    o.w = p; p += sv; o.mx = p; p += sv; o.my = p; p += sv; o.xx = p; p += sv;
    o.xy = p; p += sv; o.yy = p; p += sv; o.tr = p; p += sv; o.u = p; p += sv;
    o.alias = p;
    const u32 feat = align16u(16u * (u32)C) + align16u(8u * (u32)C) + align16u(2u * (u32)C);
    const u32 sort1 = 3u * sv;                    // the sorting network's keys and payload (the counting sort: 2 sv of keys)
    const u32 rounds = sv + PHD_RWIN_BYTES;       // two u16 lists of the unmerged survivors + the window and seed records
    u32 x = feat > sort1 ? feat : sort1;
    x = x > rounds ? x : rounds;
    const u32 tail = 4u * PHD_NW * 64u + 4u * 7u * 64u + 4u * align16u(4u * (u32)MM) + align16u(4u * (u32)(MM > 32 ? MM : 32));
    // merge_small(): rows 8 KB (+ the seed masks) U accumulators 12 KB, and — when the survivor planes are too small to hold
    // them (S < 1024) — the sorted records behind those
    const u32 small = (S >= 4 * PHD_SMALL_S) ? 48u * PHD_SMALL_S : 48u * PHD_SMALL_S + 2u * 16u * PHD_SMALL_S;
    if (x + tail < small) x = small - tail;
    p += x;
    o.part = p; p += 4u * PHD_NW * 64u;
    o.win = p; p += 4u * 7u * 64u;
    o.z_r = p; p += align16u(4u * (u32)MM);
    o.z_b = p; p += align16u(4u * (u32)MM);
    o.logZ = p; p += align16u(4u * (u32)MM);
    o.zok = p; p += align16u(4u * (u32)MM);
    o.zpart = p; p += align16u(4u * (u32)(MM > 32 ? MM : 32));
    o.wide_end = p;
    o.out_idx = p; p += align16u(2u * (u32)C);
    o.red = p; p += align16u(4u * (2 * PHD_NW + 4));
    o.ctr = p; p += 4u * 32u;
    o.total = p;
    return o;
}

} // namespace phd
