// phd_predict.h — vehicle predict (Ackerman) and the counter-based noise generator.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"
#include "phd_math.h"
#include "phd_lds.h"
#include "phd_sort.h"
#include "phd_merge.h"

namespace phd {

// ------------------------------------------------------------------------------------------
// vehicle predict (phdPredictKernelAckerman, src/phdfilter.cu:785-825) — shared by the stand-alone
// predict kernel and the fused step
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// counter-based generator: Box-Muller on two splitmix64 outputs (replaces rng.cpp's wall-clock
// seeded boost::mt19937; draw order (n_alpha, n_encoder), src/phdfilter.cu:1148-1152)
__device__ __forceinline__ void draw_noise(u64 seed, u64 counter, int i, const DevConfig& cfg, float& n_alpha,
                                           float& n_encoder)
{
    // the GLOBAL particle index: a sharded filter draws what the single filter draws
    u64 a = splitmix64(seed ^ splitmix64(counter * 0x100000001B3ull + (u64)(i + cfg.particleOffset) * 2ull));
    u64 b = splitmix64(a);
    float u1 = ((float)((a >> 40) + 1)) * (1.0f / 16777216.0f);
    float u2 = ((float)(b >> 40)) * (1.0f / 16777216.0f);
    float rad = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.2831855f * u2, &sn, &cs);
    n_alpha = cfg.stdAlpha * (rad * cs);
    n_encoder = cfg.stdEncoder * (rad * sn);
}

__device__ __forceinline__ phd_pose predict_pose(const phd_pose& o, phd_ackerman_control u, float n_alpha,
                                                 float n_encoder, const DevConfig& cfg)
{
    const float ve = u.v_encoder + n_encoder;                                   // :802
    const float al = u.alpha + n_alpha;                                         // :803
    const float tn = tanf(al);
    const float vc = ve / (1 - tn * cfg.h / cfg.l);                             // :804
    float sn, cs;
    sincosf(o.ptheta, &sn, &cs);
    const float xc_dot = vc * cs, yc_dot = vc * sn;                             // :805-806
    const float thetac_dot = vc * tn / cfg.l;                                   // :807
    const float dt = cfg.dt / cfg.subdividePredict;                             // :808
    phd_pose nw;
    nw.px = o.px + dt * (xc_dot - thetac_dot * (cfg.a * sn + cfg.b * cs));     // :809-812
    nw.py = o.py + dt * (yc_dot + thetac_dot * (cfg.a * cs - cfg.b * sn));     // :813-816
    nw.ptheta = wrap_angle(o.ptheta + dt * thetac_dot);                         // :817
    nw.vx = 0; nw.vy = 0; nw.vtheta = 0;                                        // :818-820
    return nw;
}

} // namespace phd
