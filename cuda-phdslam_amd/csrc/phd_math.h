// phd_math.h — per-feature EKF terms, Joseph-form covariance, Mahalanobis / Hellinger distances.
// Part of the one translation unit phd_kernels.hip (device code, namespace phd); see that file for the overview.
#pragma once
#include "phd_defs.h"
#include "phd_lane.h"

namespace phd {

// ------------------------------------------------------------------------------------------
// EKF terms of one in-range feature (src/phdfilter.cu:1841-1894), fp32, reference expression order
// ------------------------------------------------------------------------------------------
struct EkfTerms {
    float r, b, pd;
    float s00, s12, s11; // S = Sigma^-1: S[0], S[1]+S[2], S[3]
    float det;
    float K0, K1, K2, K3;
    float J0, J1, J2, J3;
};

__device__ __forceinline__ void ekf_terms(float mx, float my, float pxx, float pxy, float pyy,
                                          const phd_pose& pose, const DevConfig& cfg, EkfTerms& t)
{
    float dx = mx - pose.px;
    float dy = my - pose.py;
    float r2 = dx * dx + dy * dy;
    float r = sqrtf(r2);
    float bearing = wrap_angle(atan2f(dy, dx) - pose.ptheta);
    t.r = r;
    t.b = bearing;
    t.pd = (r <= cfg.maxRange && fabsf(bearing) <= cfg.maxBearing) ? cfg.pd : 0.f; // :1848-1850
    float J0 = dx / r, J2 = dy / r, J1 = -dy / r2, J3 = dx / r2;                  // :1854-1858
    const float P0 = pxx, P1 = pxy, P2 = pxy, P3 = pyy;
    float sg0 = (P0 * J0 + J2 * P1) * J0 + (J0 * P2 + P3 * J2) * J2 + cfg.stdRange * cfg.stdRange;
    float sg1 = (P0 * J1 + J3 * P1) * J0 + (J1 * P2 + P3 * J3) * J2;
    float sg2 = (P0 * J0 + J2 * P1) * J1 + (J0 * P2 + P3 * J2) * J3;
    float sg3 = (P0 * J1 + J3 * P1) * J1 + (J1 * P2 + P3 * J3) * J3 + cfg.stdBearing * cfg.stdBearing;
    sg1 = (sg1 + sg2) * 0.5f;                                                    // :1871-1872
    sg2 = sg1;
    float det = sg0 * sg3 - sg1 * sg2;                                           // :1874
    float S0 = sg3 / det, S1 = -sg1 / det, S2 = -sg2 / det, S3 = sg0 / det;      // :1877-1881
    t.det = det;
    t.s00 = S0;
    t.s12 = S1 + S2;
    t.s11 = S3;
    t.K0 = S0 * (P0 * J0 + P2 * J2) + S1 * (P0 * J1 + P2 * J3);                  // :1884-1888
    t.K1 = S0 * (P1 * J0 + P3 * J2) + S1 * (P1 * J1 + P3 * J3);
    t.K2 = S2 * (P0 * J0 + P2 * J2) + S3 * (P0 * J1 + P2 * J3);
    t.K3 = S2 * (P1 * J0 + P3 * J2) + S3 * (P1 * J1 + P3 * J3);
    t.J0 = J0; t.J1 = J1; t.J2 = J2; t.J3 = J3;
}

// Joseph-form covariance (src/phdfilter.cu:1891-1894); returns the symmetric part
__device__ __forceinline__ void joseph_cov(const EkfTerms& t, float pxx, float pxy, float pyy,
                                           const DevConfig& cfg, float& oxx, float& oxy, float& oyy)
{
    const float P0 = pxx, P1 = pxy, P2 = pxy, P3 = pyy;
    const float sr = cfg.stdRange, sb = cfg.stdBearing;
    float a00 = 1 - t.K0 * t.J0 - t.K2 * t.J1;
    float a01 = -t.K0 * t.J2 - t.K2 * t.J3;
    float a10 = -t.K1 * t.J0 - t.K3 * t.J1;
    float a11 = 1 - t.K1 * t.J2 - t.K3 * t.J3;
    float c0 = (a00 * P0 + a01 * P1) * a00 + (a00 * P2 + a01 * P3) * a01 + t.K0 * t.K0 * sr * sr + t.K2 * t.K2 * sb * sb;
    float c2 = (a00 * P0 + a01 * P1) * a10 + (a00 * P2 + a01 * P3) * a11 + t.K0 * sr * sr * t.K1 + t.K2 * sb * sb * t.K3;
    float c1 = (a10 * P0 + a11 * P1) * a00 + (a10 * P2 + a11 * P3) * a01 + t.K0 * sr * sr * t.K1 + t.K2 * sb * sb * t.K3;
    float c3 = (a10 * P0 + a11 * P1) * a10 + (a10 * P2 + a11 * P3) * a11 + t.K1 * t.K1 * sr * sr + t.K3 * t.K3 * sb * sb;
    oxx = c0;
    oxy = (c1 + c2) * 0.5f; // the merge symmetrises anyway (force_symmetric_covariance, device_math.cuh:710-725)
    oyy = c3;
}

// the birth Gaussian of a measurement (host loop src/phdfilter.cu:3470-3506): the inverse measurement model at the particle's
// pose and the measurement noise (scaled by birth_noise_factor) pushed through its Jacobian
__device__ __forceinline__ void birth_geometry(const phd_pose& pose, float zr, float zb, const DevConfig& cfg, float& mx, float& my,
                                               float& xx, float& xy, float& yy)
{
    const float theta = pose.ptheta + zb;
    float sn, cs;
    sincosf(theta, &sn, &cs);
    const float dx = zr * cs, dy = zr * sn;
    const float J0 = dx / zr, J1 = dy / zr, J2 = -dy, J3 = dx;
    const float sr = cfg.stdRange * cfg.birthNoiseFactor, sb = cfg.stdBearing * cfg.birthNoiseFactor;
    const float vr = sr * sr, vb = sb * sb;
    mx = pose.px + dx;
    my = pose.py + dy;
    xx = J0 * J0 * vr + J2 * J2 * vb;
    xy = J0 * J1 * vr + J2 * J3 * vb;
    yy = J1 * J1 * vr + J3 * J3 * vb;
}

// The detection term of in-range feature i (map slab `in`, planes of `cap`) for the innovation (i0, i1): updated mean
// (src/phdfilter.cu:1903-1904) and Joseph covariance (:1891-1894).  Gain and covariance do not depend on the measurement, but
// keeping them per feature costs 36 B of LDS each; the terms that survive the prune (a few per feature at most) rebuild them
// from the prior instead — the same two routines on the same inputs as the classification pass, hence the same bits.
// The inverse innovation covariance is NOT rebuilt: the classification pass left it in LDS for pass 1 (r, S00, S01 + S10, S11 —
// S01 = S10, so half the sum is either, exactly), which spares the innovation covariance, its determinant and four divisions.
__device__ __forceinline__ void detection_posterior(const float* __restrict__ in, int cap, int i, const phd_pose& pose,
                                                    const DevConfig& cfg, float r, float S0, float S12, float S3, float i0,
                                                    float i1, float& mx, float& my, float& xx, float& xy, float& yy)
{
    const float pmx = in[1 * cap + i], pmy = in[2 * cap + i];
    const float pxx = in[3 * cap + i], pxy = in[4 * cap + i], pyy = in[5 * cap + i];
    EkfTerms t;
    {
        // ekf_terms()'s own expressions for the Jacobian (:1854-1858) and the gain (:1884-1888), on its own values
        const float dx = pmx - pose.px, dy = pmy - pose.py;
        const float r2 = dx * dx + dy * dy;
        const float J0 = dx / r, J2 = dy / r, J1 = -dy / r2, J3 = dx / r2;
        const float P0 = pxx, P1 = pxy, P2 = pxy, P3 = pyy;
        const float S1 = 0.5f * S12, S2 = S1;
        t.K0 = S0 * (P0 * J0 + P2 * J2) + S1 * (P0 * J1 + P2 * J3);
        t.K1 = S0 * (P1 * J0 + P3 * J2) + S1 * (P1 * J1 + P3 * J3);
        t.K2 = S2 * (P0 * J0 + P2 * J2) + S3 * (P0 * J1 + P2 * J3);
        t.K3 = S2 * (P1 * J0 + P3 * J2) + S3 * (P1 * J1 + P3 * J3);
        t.J0 = J0; t.J1 = J1; t.J2 = J2; t.J3 = J3;
    }
    joseph_cov(t, pxx, pxy, pyy, cfg, xx, xy, yy);
    mx = pmx + t.K0 * i0 + t.K2 * i1;
    my = pmy + t.K1 * i0 + t.K3 * i1;
}

// ------------------------------------------------------------------------------------------
// exact distances — evaluated in the reference's operation order with FMA contraction off, so
// the merge reproduces the CPU oracle bit for bit on identical inputs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float mahal_dist(float amx, float amy, float axx, float axy, float ayy,
                                            float bmx, float bmy, float bxx, float bxy, float byy)
{
#pragma clang fp contract(off)
    // src/device_math.cuh:308-325 (+ invert_matrix2 :62-69)
    float s0 = (axx + bxx) * 0.5f;
    float s1 = (axy + bxy) * 0.5f;
    float s2 = s1;
    float s3 = (ayy + byy) * 0.5f;
    float det = s0 * s3 - s2 * s1;
    float i0v = s3 / det;
    float i1v = -s1 / det;
    float i2v = -s2 / det;
    float i3v = s0 / det;
    float d0 = amx - bmx;
    float d1 = amy - bmy;
    return d0 * d0 * i0v + d0 * d1 * (i1v + i2v) + d1 * d1 * i3v;
}

__device__ __forceinline__ float hellinger_dist(float amx, float amy, float axx, float axy, float ayy,
                                                float bmx, float bmy, float bxx, float bxy, float byy)
{
#pragma clang fp contract(off)
    // src/device_math.cuh:373-413
    float d0 = amx - bmx, d1 = amy - bmy;
    float g0 = axx + bxx, g1 = axy + bxy, g2 = g1, g3 = ayy + byy;
    float det = g0 * g3 - g2 * g1;
    float v0 = 1.f, v1 = 0.f, v2 = 0.f, v3 = 1.f;
    if (det > FLT_MIN) { v0 = g3 / det; v1 = -g1 / det; v2 = -g2 / det; v3 = g0 / det; }
    float eps = (float)(-0.25 * (double)(d0 * d0 * v0 + d0 * d1 * (v1 + v2) + d1 * d1 * v3));
    det = det / 4;
    float dist = 1 / det;
    float q0 = axx * bxx + axy * bxy;
    float q1 = axy * bxx + ayy * bxy;
    float q2 = axx * bxy + axy * byy;
    float q3 = axy * bxy + ayy * byy;
    det = q0 * q3 - q2 * q1;
    dist *= sqrtf(det);
    dist = 1 - sqrtf(dist) * expf(eps);
    return dist;
}

} // namespace phd
