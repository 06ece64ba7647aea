/*
 * scphd_cpu.c — CPU ORACLE (test infrastructure; see scphd_cpu.h for scope and pin status).
 *
 * Restates, function by function, the algorithm of cheesinglee/cuda-PHDSLAM's hot path.
 * Arithmetic is fp32 like the reference (REAL = float, src/slamtypes.h:21) and follows the
 * reference's expression order, including the places where C++ promotes to double because
 * of a double literal or M_PI.  Sums that the reference does with a thread-strided tree
 * reduction (sumByReduction, src/device_math.cuh:452-472) are done here sequentially in index
 * order: the reference's order depends on blockDim and is not a property worth pinning.
 *
 * Compile with -ffp-contract=off (see Makefile) so that the merge stage — which has no
 * transcendental in it — is reproducible bit for bit by any IEEE-754 implementation that
 * performs the same operations in the same order.
 */
#define _GNU_SOURCE
#include "scphd_cpu.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define O_LOG0 (-FLT_MAX) /* src/slamtypes.h:26 */

/* ------------------------------------------------------------------------------------ */
/* per-thread scratch (round 5).  One particle's update used a dozen malloc/free pairs —   */
/* 0.9 MB of them at 4096 x 256 x 64 — INSIDE the OpenMP loop over particles: above glibc's */
/* mmap threshold that is an mmap/munmap pair and a fresh set of page faults per particle, */
/* serialised on the process's address-space lock.  The work arrays now come from a        */
/* thread-local stack that is kept between particles (frames: o_tmp_enter / o_tmp_leave;   */
/* blocks are chained while a frame is open and merged into one when the stack empties).   */
/* Memory only: no arithmetic is touched, the outputs are bit for bit what they were.      */
/* ------------------------------------------------------------------------------------ */
#define O_TMP_BLOCKS 24
typedef struct { char* p; size_t cap, top; } o_tmp_block;
static _Thread_local o_tmp_block o_tmp_blk[O_TMP_BLOCKS];
static _Thread_local int o_tmp_cur = 0;               /* block in use */

o_tmp_frame o_tmp_enter(void)
{
    o_tmp_frame f = { o_tmp_cur, o_tmp_blk[o_tmp_cur].top };
    return f;
}

void* o_tmp_alloc(size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes == 0) bytes = 64;
    o_tmp_block* b = &o_tmp_blk[o_tmp_cur];
    if (b->p && b->top + bytes <= b->cap) { void* r = b->p + b->top; b->top += bytes; return r; }
    /* next block: at least twice everything held so far (so the chain stays short) */
    size_t held = 0;
    for (int i = 0; i <= o_tmp_cur; i++) held += o_tmp_blk[i].cap;
    int k = b->p ? o_tmp_cur + 1 : o_tmp_cur;
    if (k >= O_TMP_BLOCKS) return NULL;
    size_t want = bytes > 2 * held ? bytes : 2 * held;
    if (want < ((size_t)1 << 16)) want = (size_t)1 << 16;
    o_tmp_block* nb = &o_tmp_blk[k];
    if (nb->cap < want) {
        free(nb->p);
        nb->p = (char*)aligned_alloc(64, want);
        nb->cap = nb->p ? want : 0;
        if (!nb->p) return NULL;
    }
    nb->top = bytes;
    o_tmp_cur = k;
    return nb->p;
}

void o_tmp_leave(o_tmp_frame f)
{
    o_tmp_cur = f.block;
    o_tmp_blk[f.block].top = f.top;
    if (f.block == 0 && f.top == 0 && o_tmp_blk[1].p) {
        /* the stack is empty and it needed more than one block: keep ONE block of the total size for the next particle */
        size_t total = 0;
        for (int i = 0; i < O_TMP_BLOCKS; i++) { total += o_tmp_blk[i].cap; free(o_tmp_blk[i].p); o_tmp_blk[i].p = NULL; o_tmp_blk[i].cap = 0; o_tmp_blk[i].top = 0; }
        o_tmp_blk[0].p = (char*)aligned_alloc(64, total);
        o_tmp_blk[0].cap = o_tmp_blk[0].p ? total : 0;
    }
}

/* give the calling thread's scratch back (a long-lived host that is done with the oracle) */
void o_tmp_release(void)
{
    for (int i = 0; i < O_TMP_BLOCKS; i++) { free(o_tmp_blk[i].p); o_tmp_blk[i].p = NULL; o_tmp_blk[i].cap = 0; o_tmp_blk[i].top = 0; }
    o_tmp_cur = 0;
}

int o_omp_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* src/device_math.cuh:9-16 */
float o_safe_log(float x) { return (x <= 0) ? O_LOG0 : logf(x); }

/* src/device_math.cuh:241-251.  M_PI is a double there: the comparisons and the +-2*pi are
 * evaluated in double and rounded back to float. */
float o_wrap_angle(float a)
{
    float remainder = fmodf(a, (float)(2 * M_PI));
    if ((double)remainder > M_PI)
        remainder = (float)((double)remainder - 2 * M_PI);
    else if ((double)remainder < -M_PI)
        remainder = (float)((double)remainder + 2 * M_PI);
    return remainder;
}

/*
 * Portable exponential used for the resampling CDF.  The reference calls libm exp() on the
 * float log-weight and accumulates in double (src/main.cpp:463,495); libm's last-bit rounding
 * is platform specific, which would make "resampling indices bit-exact" a matter of luck.
 * This routine uses only IEEE-754 basic operations (mul, fma, rint, ldexp), so every
 * conforming CPU and GPU produces the same double.  |error| < 1 ulp(double).
 */
double o_det_exp(float xf)
{
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double x = (double)xf;
    if (!(x >= -700.0)) return (x != x) ? x : 0.0;
    if (x > 700.0) return INFINITY;
    double kd = rint(x * LOG2E);
    double r = fma(-kd, LN2_HI, x);
    r = fma(-kd, LN2_LO, r);
    /* Taylor, degree 13, |r| <= 0.3466 */
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}

/* ------------------------------------------------------------------------------------ */
/* predict: phdPredictKernelAckerman, src/phdfilter.cu:785-825 (nPredictParticles = 1)   */
/* ------------------------------------------------------------------------------------ */
void o_predict_ackerman(o_pose* poses, int n, float alpha, float v_encoder,
                        const float* noise, const o_config* cfg)
{
    for (int i = 0; i < n; i++) {
        o_pose old = poses[i];
        float n_alpha = noise ? noise[2 * i] : 0.0f;
        float n_encoder = noise ? noise[2 * i + 1] : 0.0f;
        float ve_noisy = v_encoder + n_encoder;                       /* :802 */
        float alpha_noisy = alpha + n_alpha;                          /* :803 */
        float vc = ve_noisy / (1 - tanf(alpha_noisy) * cfg->h / cfg->l); /* :804 */
        float xc_dot = vc * cosf(old.ptheta);                         /* :805 */
        float yc_dot = vc * sinf(old.ptheta);                         /* :806 */
        float thetac_dot = vc * tanf(alpha_noisy) / cfg->l;           /* :807 */
        float dt = cfg->dt / cfg->subdividePredict;                   /* :808 */
        o_pose nw;
        nw.px = old.px + dt * (xc_dot - thetac_dot * (cfg->a * sinf(old.ptheta) + cfg->b * cosf(old.ptheta)));
        nw.py = old.py + dt * (yc_dot + thetac_dot * (cfg->a * cosf(old.ptheta) - cfg->b * sinf(old.ptheta)));
        nw.ptheta = o_wrap_angle(old.ptheta + dt * thetac_dot);       /* :817 */
        nw.vx = 0; nw.vy = 0; nw.vtheta = 0;                          /* :818-820 */
        poses[i] = nw;
    }
}

/* predicted range/bearing of a feature mean seen from a pose: the expression shared by
 * computeInRangeKernel (src/phdfilter.cu:1328-1332) and preUpdateSynthKernel (:1841-1845) */
void o_predicted_measurement(const o_pose* pose, const float* mean, float* r_out, float* r2_out,
                             float* b_out, float* dx_out, float* dy_out)
{
    float dx = mean[0] - pose->px;
    float dy = mean[1] - pose->py;
    float r2 = dx * dx + dy * dy;
    float r = sqrtf(r2);
    float bearing = o_wrap_angle(atan2f(dy, dx) - pose->ptheta);
    *r_out = r; *b_out = bearing;
    if (r2_out) *r2_out = r2;
    if (dx_out) *dx_out = dx;
    if (dy_out) *dy_out = dy;
}

/* ------------------------------------------------------------------------------------ */
/* in-range classification: computeInRangeKernel, src/phdfilter.cu:1328-1346              */
/* ------------------------------------------------------------------------------------ */
void o_classify(const o_gaussian* map, int n, const o_pose* pose, const o_config* cfg, int8_t* cls)
{
    for (int i = 0; i < n; i++) {
        float r, bearing;
        o_predicted_measurement(pose, map[i].mean, &r, NULL, &bearing, NULL, NULL);
        cls[i] = 0;
        if (r >= cfg->minRange && r <= cfg->maxRange && fabsf(bearing) <= cfg->maxBearing)
            cls[i] = 1;
        /* the 0.8 / 1.2 literals are doubles: comparison in double (:1340-1342) */
        else if ((double)r >= 0.8 * (double)cfg->minRange && (double)r <= 1.2 * (double)cfg->maxRange &&
                 (double)fabsf(bearing) <= 1.2 * (double)cfg->maxBearing)
            cls[i] = 2;
    }
}

/* ------------------------------------------------------------------------------------ */
/* births: host loop src/phdfilter.cu:3470-3506                                           */
/* (cos/sin of a float stay float under <cmath>)                                          */
/* ------------------------------------------------------------------------------------ */
void o_births(const o_pose* pose, const o_meas* z, int M, const o_config* cfg, o_gaussian* births)
{
    for (int j = 0; j < M; j++) {
        float theta = pose->ptheta + z[j].bearing;
        float dx = z[j].range * cosf(theta);
        float dy = z[j].range * sinf(theta);
        births[j].mean[0] = pose->px + dx;
        births[j].mean[1] = pose->py + dy;
        float J[4];
        J[0] = dx / z[j].range;
        J[1] = dy / z[j].range;
        J[2] = -dy;
        J[3] = dx;
        /* pow(x,2) with a float x: CUDA's pow(float,int) overload is a float multiply */
        float sr = cfg->stdRange * cfg->birthNoiseFactor;
        float sb = cfg->stdBearing * cfg->birthNoiseFactor;
        float var_range = sr * sr;
        float var_bearing = sb * sb;
        births[j].cov[0] = J[0] * J[0] * var_range + J[2] * J[2] * var_bearing;
        births[j].cov[1] = J[0] * J[1] * var_range + J[2] * J[3] * var_bearing;
        births[j].cov[2] = births[j].cov[1];
        births[j].cov[3] = J[1] * J[1] * var_range + J[3] * J[3] * var_bearing;
        if (z[j].label == 0 || !cfg->labeledMeasurements)
            births[j].weight = o_safe_log(cfg->birthWeight);
        else
            births[j].weight = o_safe_log(0);
    }
}

/* ------------------------------------------------------------------------------------ */
/* EKF pre-update: preUpdateSynthKernel (2D), src/phdfilter.cu:1833-1924                   */
/* ------------------------------------------------------------------------------------ */
void o_preupdate(const o_pose* pose, const o_gaussian* feat, int n, const o_meas* z, int M,
                 const o_config* cfg, float* pd, o_gaussian* preupdate)
{
    for (int i = 0; i < n; i++) {
        const o_gaussian* f = &feat[i];
        float dx, dy, r2, r, bearing;
        o_predicted_measurement(pose, f->mean, &r, &r2, &bearing, &dx, &dy);

        float feature_pd = 0;                                          /* :1848-1851 */
        if (r <= cfg->maxRange && fabsf(bearing) <= cfg->maxBearing) feature_pd = cfg->pd;
        pd[i] = feature_pd;

        float J[4];                                                    /* :1854-1858 */
        J[0] = dx / r;
        J[2] = dy / r;
        J[1] = -dy / r2;
        J[3] = dx / r2;
        const float* P = f->cov;

        /* pow(stdRange,2) is CUDA's pow(float,int) overload: a float multiply (:1865,1868) */
        float sigma[4];
        sigma[0] = (P[0] * J[0] + J[2] * P[1]) * J[0] + (J[0] * P[2] + P[3] * J[2]) * J[2] + cfg->stdRange * cfg->stdRange;
        sigma[1] = (P[0] * J[1] + J[3] * P[1]) * J[0] + (J[1] * P[2] + P[3] * J[3]) * J[2];
        sigma[2] = (P[0] * J[0] + J[2] * P[1]) * J[1] + (J[0] * P[2] + P[3] * J[2]) * J[3];
        sigma[3] = (P[0] * J[1] + J[3] * P[1]) * J[1] + (J[1] * P[2] + P[3] * J[3]) * J[3] + cfg->stdBearing * cfg->stdBearing;
        sigma[1] = (sigma[1] + sigma[2]) / 2;                           /* :1871-1872 */
        sigma[2] = sigma[1];
        float det_sigma = sigma[0] * sigma[3] - sigma[1] * sigma[2];    /* :1874 */
        float S[4];                                                    /* :1877-1881 */
        S[0] = sigma[3] / det_sigma;
        S[1] = -sigma[1] / det_sigma;
        S[2] = -sigma[2] / det_sigma;
        S[3] = sigma[0] / det_sigma;
        float K[4];                                                    /* :1884-1888 */
        K[0] = S[0] * (P[0] * J[0] + P[2] * J[2]) + S[1] * (P[0] * J[1] + P[2] * J[3]);
        K[1] = S[0] * (P[1] * J[0] + P[3] * J[2]) + S[1] * (P[1] * J[1] + P[3] * J[3]);
        K[2] = S[2] * (P[0] * J[0] + P[2] * J[2]) + S[3] * (P[0] * J[1] + P[2] * J[3]);
        K[3] = S[2] * (P[1] * J[0] + P[3] * J[2]) + S[3] * (P[1] * J[1] + P[3] * J[3]);

        /* Joseph form (I-KH)P(I-KH)^T + K R K^T, :1891-1894, same association as the reference */
        float sr = cfg->stdRange, sb = cfg->stdBearing;
        float a00 = 1 - K[0] * J[0] - K[2] * J[1];
        float a01 = -K[0] * J[2] - K[2] * J[3];
        float a10 = -K[1] * J[0] - K[3] * J[1];
        float a11 = 1 - K[1] * J[2] - K[3] * J[3];
        float cu[4];
        cu[0] = (a00 * P[0] + a01 * P[1]) * a00 + (a00 * P[2] + a01 * P[3]) * a01 + K[0] * K[0] * sr * sr + K[2] * K[2] * sb * sb;
        cu[2] = (a00 * P[0] + a01 * P[1]) * a10 + (a00 * P[2] + a01 * P[3]) * a11 + K[0] * sr * sr * K[1] + K[2] * sb * sb * K[3];
        cu[1] = (a10 * P[0] + a11 * P[1]) * a00 + (a10 * P[2] + a11 * P[3]) * a01 + K[0] * sr * sr * K[1] + K[2] * sb * sb * K[3];
        cu[3] = (a10 * P[0] + a11 * P[1]) * a10 + (a10 * P[2] + a11 * P[3]) * a11 + K[1] * K[1] * sr * sr + K[3] * K[3] * sb * sb;

        float log_2pi = o_safe_log((float)(2 * M_PI));
        float log_det = o_safe_log(det_sigma);
        for (int m = 0; m < M; m++) {                                  /* :1898-1923 */
            o_gaussian* g = &preupdate[(size_t)m * n + i];
            float innov0 = z[m].range - r;
            float innov1 = o_wrap_angle(z[m].bearing - bearing);
            g->mean[0] = f->mean[0] + K[0] * innov0 + K[2] * innov1;
            g->mean[1] = f->mean[1] + K[1] * innov0 + K[3] * innov1;
            for (int k = 0; k < 4; k++) g->cov[k] = cu[k];
            float dist = innov0 * innov0 * S[0] + innov0 * innov1 * (S[1] + S[2]) + innov1 * innov1 * S[3];
            /* "- 0.5*dist - safeLog(2*M_PI) - 0.5*safeLog(det_sigma)": 0.5 is a double (:1911) */
            float gl = (float)(-0.5 * (double)dist - (double)log_2pi - 0.5 * (double)log_det);
            if (z[m].label == 0 || !cfg->labeledMeasurements)
                g->weight = o_safe_log(feature_pd) + o_safe_log(f->weight) + gl; /* :1916-1917 */
            else
                g->weight = o_safe_log(0);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* GM-PHD update: phdUpdateKernel, src/phdfilter.cu:2119-2319 (particle_weighting == 0)    */
/* ------------------------------------------------------------------------------------ */
void o_update(const o_gaussian* feat, const float* pd, const o_gaussian* preupdate,
              const o_gaussian* births, int n, int M, const o_config* cfg,
              o_gaussian* slab, uint8_t* prune_flag, float* dlogw)
{
    int n_update = n * (M + 1) + M;
    float cardinality_predict = 0;
    /* slab order :2145-2172: [non-detect | detect m-major | births] */
    for (int j = 0; j < n; j++) {
        slab[j] = feat[j];
        slab[j].weight = feat[j].weight * (1 - pd[j]);                  /* :2148 */
        for (int m = 0; m < M; m++) slab[n + (size_t)m * n + j] = preupdate[(size_t)m * n + j];
        cardinality_predict += pd[j] * feat[j].weight;                  /* :2160,2183-2184 */
    }
    for (int m = 0; m < M; m++) {
        slab[n + (size_t)M * n + m] = births[m];
        cardinality_predict += cfg->birthWeight;                        /* :2174 */
    }
    float particle_weight = 0;
    for (int m = 0; m < M; m++) {                                       /* :2190-2253 */
        o_gaussian* ptr = slab + n + (size_t)m * n;
        float sum = 0;
        for (int j = 0; j < n; j++) sum += expf(ptr[j].weight);         /* :2205-2209 */
        sum += cfg->clutterDensity;                                      /* :2213 */
        sum += cfg->birthWeight;                                         /* :2214 */
        float log_normalizer = o_safe_log(sum);                          /* :2217 */
        for (int j = 0; j < n; j++) ptr[j].weight = expf(ptr[j].weight - log_normalizer); /* :2242-2243 */
        o_gaussian* b = slab + n + (size_t)M * n + m;                    /* :2239 */
        b->weight = expf(b->weight - log_normalizer);
        particle_weight += log_normalizer;                               /* :2251 */
    }
    particle_weight -= cardinality_predict;                              /* :2261 */
    *dlogw = particle_weight;
    for (int i = 0; i < n_update; i++)                                   /* :2308-2318 */
        prune_flag[i] = (slab[i].weight < cfg->minFeatureWeight) ? 1 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* distances: src/device_math.cuh:62-69,308-325 and :373-413                               */
/* ------------------------------------------------------------------------------------ */
float o_mahal_dist(const o_gaussian* a, const o_gaussian* b)
{
    float sigma[4], inv[4];
    for (int i = 0; i < 4; i++) sigma[i] = (a->cov[i] + b->cov[i]) / 2;
    float det = sigma[0] * sigma[3] - sigma[2] * sigma[1];
    inv[0] = sigma[3] / det;
    inv[1] = -sigma[1] / det;
    inv[2] = -sigma[2] / det;
    inv[3] = sigma[0] / det;
    float i0 = a->mean[0] - b->mean[0];
    float i1 = a->mean[1] - b->mean[1];
    return i0 * i0 * inv[0] + i0 * i1 * (inv[1] + inv[2]) + i1 * i1 * inv[3];
}

float o_hellinger_dist(const o_gaussian* a, const o_gaussian* b)
{
    float innov0 = a->mean[0] - b->mean[0];
    float innov1 = a->mean[1] - b->mean[1];
    float sigma[4], sinv[4] = {1, 0, 0, 1};
    for (int i = 0; i < 4; i++) sigma[i] = a->cov[i] + b->cov[i];
    float det = sigma[0] * sigma[3] - sigma[2] * sigma[1];
    if (det > FLT_MIN) {
        sinv[0] = sigma[3] / det;
        sinv[1] = -sigma[1] / det;
        sinv[2] = -sigma[2] / det;
        sinv[3] = sigma[0] / det;
    }
    /* -0.25 is a double literal (:394) */
    float epsilon = (float)(-0.25 * (double)(innov0 * innov0 * sinv[0] + innov0 * innov1 * (sinv[1] + sinv[2]) +
                                              innov1 * innov1 * sinv[3]));
    det /= 4;
    float dist = 1 / det;
    sigma[0] = a->cov[0] * b->cov[0] + a->cov[2] * b->cov[1];
    sigma[1] = a->cov[1] * b->cov[0] + a->cov[3] * b->cov[1];
    sigma[2] = a->cov[0] * b->cov[2] + a->cov[2] * b->cov[3];
    sigma[3] = a->cov[1] * b->cov[2] + a->cov[3] * b->cov[3];
    det = sigma[0] * sigma[3] - sigma[2] * sigma[1];
    dist *= sqrtf(det);
    dist = 1 - sqrtf(dist) * expf(epsilon);
    return dist;
}

/* ------------------------------------------------------------------------------------ */
/* merge: phdUpdateMergeKernel, src/phdfilter.cu:2739-2890                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct { float w; int idx; } o_sortkey;
static int o_cmp_desc(const void* pa, const void* pb)
{
    const o_sortkey* a = (const o_sortkey*)pa;
    const o_sortkey* b = (const o_sortkey*)pb;
    if (a->w > b->w) return -1;
    if (a->w < b->w) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* ------------------------------------------------------------------------------------ */
/* Exact, order-free moment sums (o_config.mergeSums == 0).
 *
 * The reference adds the members of a cluster with sumByReduction (src/phdfilter.cu:2795-2881): per-thread partial sums
 * over a strided slice, then a tree — an order that depends on the block size.  A float sum has no value of its own
 * under such a rule, an integer sum has.  So each TERM is the reference's float expression (one rounding per operation,
 * no contraction), and the terms are added as integers after an exact conversion (a mantissa shift) to a fixed-point
 * scale anchored at the cluster's own magnitudes:
 *
 *   field(x) = biased exponent of x, 1 for zero/denormals: |x| < 2^(field - 126)
 *   fix(x, F, top) = sign(x) * floor(|x| * 2^(top + 150 - F))          requires field(x) <= F, x finite
 *
 *   weights        q = fix(w_i, Fw, 16),  Fw = field(weight of the seed)                 (the seed is the heaviest member)
 *   weighted means q = fix(w_i * m_i, Fw + 28, 38), added as two 31-bit digits           (:2813)
 *   covariances    q = fix(w_i * (P_i + d d^T), Fc, 18),  d = mean - m_i                 (:2854-2866)
 *                  Fc = max(1, Fw + ec - 121),  ec = largest field among the members' P entries and squared
 *                  offsets (m_i - m_seed)^2 — the merged mean is within twice the largest offset of every member
 *
 * W = sum(w) and the three quotients of :2828 / :2879 are then taken in double from the integer sums (exact below 2^53)
 * and rounded to float once.  A term that violates its anchor or is not finite poisons the cluster (NaN output).
 * The device (cuda-phdslam_amd/csrc/phd_fixsum.h) implements the same definition with LDS atomics. */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int64_t W, xh, xl, yh, yl, cxx, cxy, cyy;
    int Fw, ec, ok;
    float smx, smy;
} o_exact_sums;

static int o_field(float x)
{
    uint32_t b; memcpy(&b, &x, 4);
    int ef = (int)((b >> 23) & 0xFF);
    return ef ? ef : 1;
}

static int64_t o_fix(float x, int F, int top, int* ok)
{
    uint32_t b; memcpy(&b, &x, 4);
    int ef = (int)((b >> 23) & 0xFF);
    uint32_t frac = b & 0x7FFFFFu;
    if (ef == 255) { *ok = 0; return 0; }
    uint64_t m = ef ? (uint64_t)(frac | 0x800000u) : (uint64_t)frac;
    int e1 = ef ? ef : 1;
    if (e1 > F) { *ok = 0; return 0; }
    int k = e1 - F + top;
    uint64_t q = (k >= 0) ? (m << k) : (k > -32 ? (m >> (-k)) : 0);
    return (b >> 31) ? -(int64_t)q : (int64_t)q;
}

static void o_exact_begin(o_exact_sums* s, const o_gaussian* seed)
{
    memset(s, 0, sizeof(*s));
    s->ok = 1;
    s->Fw = o_field(seed->weight);
    s->smx = seed->mean[0]; s->smy = seed->mean[1];
}

static void o_exact_digits(int64_t q, int64_t* hi, int64_t* lo)
{
    uint64_t a = q < 0 ? (uint64_t)(-q) : (uint64_t)q;
    int64_t h = (int64_t)(a >> 31), l = (int64_t)(a & 0x7FFFFFFFu);
    *hi += q < 0 ? -h : h;
    *lo += q < 0 ? -l : l;
}

static void o_exact_first(o_exact_sums* s, const o_gaussian* g)
{
    const float w = g->weight;
    s->W += o_fix(w, s->Fw, 16, &s->ok);
    const float tx = w * g->mean[0], ty = w * g->mean[1];                 /* :2813 */
    o_exact_digits(o_fix(tx, s->Fw + 28, 38, &s->ok), &s->xh, &s->xl);
    o_exact_digits(o_fix(ty, s->Fw + 28, 38, &s->ok), &s->yh, &s->yl);
    const float dx = g->mean[0] - s->smx, dy = g->mean[1] - s->smy;
    const float dx2 = dx * dx, dy2 = dy * dy;
    /* our maps are symmetric: cov[1] == cov[2] (force_symmetric_covariance, src/device_math.cuh:710-725) */
    const float v[5] = {g->cov[0], g->cov[1], g->cov[3], dx2, dy2};
    for (int k = 0; k < 5; k++) {
        uint32_t b; memcpy(&b, &v[k], 4);
        int ef = (int)((b >> 23) & 0xFF);
        int e = ef ? ef : 1;                                             /* 255 for a non-finite value */
        if (e > s->ec) s->ec = e;
    }
}

static void o_exact_mean(const o_exact_sums* s, o_gaussian* mg)
{
    const double Wd = (double)s->W;
    mg->weight = (float)ldexp(Wd, s->Fw - 166);
    const double sx = fma((double)s->xh, 2147483648.0, (double)s->xl);
    const double sy = fma((double)s->yh, 2147483648.0, (double)s->yl);
    mg->mean[0] = (float)(sx / Wd * 64.0);                               /* :2828; 2^(Fw+28-38-150) / 2^(Fw-16-150) = 2^6 */
    mg->mean[1] = (float)(sy / Wd * 64.0);
}

static int o_exact_anchor(const o_exact_sums* s)
{
    int F = s->Fw + s->ec - 121;
    return F < 1 ? 1 : F;
}

static void o_exact_second(o_exact_sums* s, const o_gaussian* mg, const o_gaussian* g)
{
    const int Fc = o_exact_anchor(s);
    const float d0 = mg->mean[0] - g->mean[0], d1 = mg->mean[1] - g->mean[1];   /* :2854-2855 */
    const float txx = g->weight * (g->cov[0] + d0 * d0);                      /* :2863-2866 */
    const float txy = g->weight * (g->cov[1] + d0 * d1);
    const float tyy = g->weight * (g->cov[3] + d1 * d1);
    s->cxx += o_fix(txx, Fc, 18, &s->ok);
    s->cxy += o_fix(txy, Fc, 18, &s->ok);
    s->cyy += o_fix(tyy, Fc, 18, &s->ok);
}

static void o_exact_cov(const o_exact_sums* s, o_gaussian* mg)
{
    if (!s->ok || s->ec >= 255) {
        mg->weight = mg->mean[0] = mg->mean[1] = NAN;
        for (int j = 0; j < 4; j++) mg->cov[j] = NAN;
        return;
    }
    const int Fc = o_exact_anchor(s);
    const double Wd = (double)s->W;
    const int sh = (Fc - 18) - (s->Fw - 16);
    mg->cov[0] = (float)ldexp((double)s->cxx / Wd, sh);                   /* :2879 */
    mg->cov[1] = (float)ldexp((double)s->cxy / Wd, sh);
    mg->cov[2] = mg->cov[1];
    mg->cov[3] = (float)ldexp((double)s->cyy / Wd, sh);
}

int o_merge(const o_gaussian* in, int n, const o_config* cfg, o_gaussian* out, float* margin_out)
{
    float margin_d = FLT_MAX, margin_w = FLT_MAX;
    int n_out = 0;
    if (n <= 0) {
        if (margin_out) { margin_out[0] = margin_d; margin_out[1] = margin_w; }
        return 0;
    }
    o_tmp_frame tmp_frame = o_tmp_enter();
    o_sortkey* order = (o_sortkey*)o_tmp_alloc(sizeof(o_sortkey) * n);
    uint8_t* merged = (uint8_t*)o_tmp_alloc(n);
    uint8_t* member = (uint8_t*)o_tmp_alloc(n);
    memset(merged, 0, n);
    for (int i = 0; i < n; i++) { order[i].w = in[i].weight; order[i].idx = i; }
    /* the arg-max of :2750-2788 with ties broken towards the lowest index == walking the
     * components in (weight desc, index asc) order and taking the first unmerged one */
    qsort(order, n, sizeof(o_sortkey), o_cmp_desc);
    int first = 0;
    while (1) {
        while (first < n && merged[order[first].idx]) first++;
        if (first >= n) break;                                          /* :2784 */
        const o_gaussian* seed = &in[order[first].idx];
        /* weight gap to the next unmerged candidate (tie-break sensitivity) */
        for (int k = first + 1; k < n; k++) {
            if (!merged[order[k].idx]) {
                float gap = (seed->weight - in[order[k].idx].weight) / (fabsf(seed->weight) + FLT_MIN);
                if (gap < margin_w) margin_w = gap;
                break;
            }
        }
        /* pass 1 (:2795-2830): members, weight, weighted mean — in sorted order, seed first */
        float W = 0, sx = 0, sy = 0;
        o_exact_sums es;
        o_exact_begin(&es, seed);
        for (int k = first; k < n; k++) {
            int i = order[k].idx;
            member[i] = 0;
            if (merged[i]) continue;
            float d = (cfg->distanceMetric == 0) ? o_mahal_dist(seed, &in[i]) : o_hellinger_dist(seed, &in[i]);
            if (k != first) {
                float mg = fabsf(d - cfg->minSeparation) / (fabsf(cfg->minSeparation) + FLT_MIN);
                if (mg < margin_d) margin_d = mg;
            }
            /* a NaN distance (Hellinger metric on a near-singular covariance: sqrt of a cancelled, slightly
             * negative determinant, src/device_math.cuh:403-408) decides by the rounding noise of its inputs */
            if (d != d) margin_d = 0;
            if (d < cfg->minSeparation) {                                /* :2806 */
                member[i] = 1;
                W += in[i].weight;
                sx += in[i].weight * in[i].mean[0];
                sy += in[i].weight * in[i].mean[1];
                o_exact_first(&es, &in[i]);
            }
        }
        o_gaussian mg;
        if (cfg->mergeSums == 1) {
            if (W == 0) break;                                          /* :2821 */
            mg.weight = W;
            mg.mean[0] = sx / W;                                        /* :2828 */
            mg.mean[1] = sy / W;
        } else {
            if (es.ok && es.ec < 255 && es.W == 0) break;               /* :2821 on the exact sum */
            o_exact_mean(&es, &mg);
        }
        /* pass 2 (:2837-2881) */
        float c[4] = {0, 0, 0, 0};
        for (int k = first; k < n; k++) {
            int i = order[k].idx;
            if (merged[i] || !member[i]) continue;
            float d0 = mg.mean[0] - in[i].mean[0];                       /* :2854-2855 */
            float d1 = mg.mean[1] - in[i].mean[1];
            float dd[2] = {d0, d1};
            for (int j = 0; j < 2; j++)
                for (int kk = 0; kk < 2; kk++)
                    c[j * 2 + kk] += in[i].weight * (in[i].cov[j * 2 + kk] + dd[j] * dd[kk]); /* :2863-2866 */
            if (cfg->mergeSums != 1) o_exact_second(&es, &mg, &in[i]);
            merged[i] = 1;                                              /* :2869 */
        }
        if (cfg->mergeSums == 1) {
            for (int j = 0; j < 4; j++) mg.cov[j] = c[j] / W;            /* :2879 */
            /* force_symmetric_covariance, src/device_math.cuh:710-725: lower = (lower+upper)/2 */
            mg.cov[1] = (mg.cov[1] + mg.cov[2]) / 2;
            mg.cov[2] = mg.cov[1];
        } else {
            o_exact_cov(&es, &mg);
        }
        out[n_out++] = mg;                                              /* :2885-2887 */
    }
    o_tmp_leave(tmp_frame);
    if (margin_out) { margin_out[0] = margin_d; margin_out[1] = margin_w; }
    return n_out;
}

/* ------------------------------------------------------------------------------------ */
/* o_merge_follow — TEST DIAGNOSTIC (no counterpart in the reference).
 *
 * The device's merge is bit for bit o_merge() of the device's own survivors `ref`.  Those survivors equal the oracle's
 * `in` only within the update stage's tolerances (libm vs the device library), so a decision of the greedy loop — is this
 * candidate within minSeparation of the seed (src/phdfilter.cu:2806), which unmerged component is the heaviest (:2750-2788) —
 * that sits within that difference of its threshold may fall the other way, and every later cluster changes with it.
 * This routine PROVES such a flip instead of tolerating it: it runs the merge of `in` but takes every decision from `ref`,
 * and for every decision where `in` alone would have decided otherwise it checks that the survivor difference accounts for it:
 *
 *   distance:  d(ref) < T but d(in) >= T (or the reverse).  With g = the gradient of the distance in its ten arguments
 *              (both means, both symmetric covariances; central differences in double at `ref`), the first-order change
 *              of d under the observed difference is at most B1 = sum_k |g_k| |in_k - ref_k|.  The flip is explained iff
 *              |d64(in) - T| <= kappa B1 + |d32(in) - d64(in)| + |d32(ref) - d64(ref)|      (kappa = 2: second-order terms;
 *              the last two terms are the float evaluation errors of the two distances, measured, not modelled).
 *   order:     the seed taken from ref's (weight desc, index asc) order is lighter in `in` than a still unmerged candidate j:
 *              explained iff w_in[j] - w_in[s] <= kappa (|w_in[j] - w_ref[j]| + |w_in[s] - w_ref[s]|).
 *
 * Output: the merged map of `in` under ref's decisions (moment sums per cfg->mergeSums, in ref's order) — to be compared,
 * component by component and IN ORDER, with the device's map — and
 *   stats[0] distance flips followed      stats[1] of them unexplained      stats[2] worst |d64(in) - T| / allowance over the flips
 *   stats[3] order inversions followed    stats[4] of them unexplained      stats[5] worst gap / allowance over the inversions
 *   stats[6] largest |d32(in) - d32(ref)| / T over the decisions with a distance below 2 T (how far the survivor difference
 *            moves a distance that matters)
 *   stats[7] number of distance decisions taken
 *   stats[8] decisions where a distance is NaN on either side (Hellinger metric on a cancelled determinant,
 *            src/device_math.cuh:403-408: decided by rounding noise — counted, never "explained")                         */
/* ------------------------------------------------------------------------------------ */
static double o_dist64(int metric, const double* x)
{
    /* x = (a.mx, a.my, a.cxx, a.cxy, a.cyy, b.mx, b.my, b.cxx, b.cxy, b.cyy) */
    const double d0 = x[0] - x[5], d1 = x[1] - x[6];
    if (metric == 0) {
        const double s0 = 0.5 * (x[2] + x[7]), s1 = 0.5 * (x[3] + x[8]), s3 = 0.5 * (x[4] + x[9]);
        const double det = s0 * s3 - s1 * s1;
        return (d0 * d0 * s3 - 2.0 * d0 * d1 * s1 + d1 * d1 * s0) / det;
    }
    const double s0 = x[2] + x[7], s1 = x[3] + x[8], s3 = x[4] + x[9];
    const double det = s0 * s3 - s1 * s1;
    const double eps = -0.25 * (d0 * d0 * s3 - 2.0 * d0 * d1 * s1 + d1 * d1 * s0) / det;
    const double da = x[2] * x[4] - x[3] * x[3], db = x[7] * x[9] - x[8] * x[8];
    return 1.0 - sqrt(sqrt(da * db) / (det / 4.0)) * exp(eps);
}

static void o_pack10(const o_gaussian* a, const o_gaussian* b, double* x)
{
    x[0] = a->mean[0]; x[1] = a->mean[1]; x[2] = a->cov[0]; x[3] = a->cov[1]; x[4] = a->cov[3];
    x[5] = b->mean[0]; x[6] = b->mean[1]; x[7] = b->cov[0]; x[8] = b->cov[1]; x[9] = b->cov[3];
}

int o_merge_follow(const o_gaussian* ref, const o_gaussian* in, int n, const o_config* cfg, o_gaussian* out, double* stats)
{
    const double kappa = 2.0;
    const float T = cfg->minSeparation;
    for (int k = 0; k < 9; k++) stats[k] = 0;
    if (n <= 0) return 0;
    int n_out = 0;
    o_tmp_frame tmp_frame = o_tmp_enter();
    o_sortkey* order = (o_sortkey*)o_tmp_alloc(sizeof(o_sortkey) * n);
    uint8_t* merged = (uint8_t*)o_tmp_alloc(n);
    uint8_t* member = (uint8_t*)o_tmp_alloc(n);
    memset(merged, 0, n);
    for (int i = 0; i < n; i++) { order[i].w = ref[i].weight; order[i].idx = i; }
    qsort(order, n, sizeof(o_sortkey), o_cmp_desc);
    int first = 0;
    while (1) {
        while (first < n && merged[order[first].idx]) first++;
        if (first >= n) break;
        const int s = order[first].idx;
        /* would `in` have picked another seed?  (its own order: weight desc, index asc; one inversion per seed, judged on the
         * candidate that is hardest to explain) */
        {
            int inv = 0, bad = 0;
            for (int k = first + 1; k < n; k++) {
                const int j = order[k].idx;
                if (merged[j]) continue;
                const int heavier = in[j].weight > in[s].weight || (in[j].weight == in[s].weight && j < s);
                if (!heavier) continue;
                const double gap = (double)in[j].weight - (double)in[s].weight;
                const double allow = kappa * (fabs((double)in[j].weight - (double)ref[j].weight) + fabs((double)in[s].weight - (double)ref[s].weight));
                const double ratio = allow > 0 ? gap / allow : (gap > 0 ? INFINITY : 0.0);
                inv = 1;
                if (ratio > stats[5]) stats[5] = ratio;
                if (gap > allow) bad = 1;
            }
            stats[3] += inv; stats[4] += bad;
        }
        float W = 0, sx = 0, sy = 0;
        o_exact_sums es;
        o_exact_begin(&es, &in[s]);
        for (int k = first; k < n; k++) {
            const int i = order[k].idx;
            member[i] = 0;
            if (merged[i]) continue;
            const float dr = (cfg->distanceMetric == 0) ? o_mahal_dist(&ref[s], &ref[i]) : o_hellinger_dist(&ref[s], &ref[i]);
            const float di = (cfg->distanceMetric == 0) ? o_mahal_dist(&in[s], &in[i]) : o_hellinger_dist(&in[s], &in[i]);
            const int take = dr < T;                                     /* the device's decision */
            if (k != first) {
                stats[7] += 1;
                const double mv = fabs((double)di - (double)dr) / (fabs((double)T) + FLT_MIN);
                if (mv == mv && mv > stats[6] && (di < 2 * T || dr < 2 * T)) stats[6] = mv;
            }
            if (di != di || dr != dr) stats[8] += 1;
            else if ((di < T) != take) {
                double xr[10], xi[10], B1 = 0;
                o_pack10(&ref[s], &ref[i], xr);
                o_pack10(&in[s], &in[i], xi);
                for (int q = 0; q < 10; q++) {
                    const double dq = fabs(xi[q] - xr[q]);
                    if (dq == 0) continue;
                    double xp[10], xm[10];
                    memcpy(xp, xr, sizeof xp); memcpy(xm, xr, sizeof xm);
                    const double h = 1e-6 * (fabs(xr[q]) > 1e-3 ? fabs(xr[q]) : 1e-3);
                    xp[q] += h; xm[q] -= h;
                    const double g = (o_dist64(cfg->distanceMetric, xp) - o_dist64(cfg->distanceMetric, xm)) / (2 * h);
                    B1 += fabs(g) * dq;
                }
                const double d64i = o_dist64(cfg->distanceMetric, xi), d64r = o_dist64(cfg->distanceMetric, xr);
                const double allow = kappa * B1 + fabs((double)di - d64i) + fabs((double)dr - d64r);
                const double off = fabs(d64i - (double)T);
                stats[0] += 1;
                /* a NaN anywhere (singular covariances under the Hellinger metric) explains nothing */
                const double ratio = (allow > 0 && off == off && allow == allow) ? off / allow : INFINITY;
                if (ratio > stats[2]) stats[2] = ratio;
                if (!(off <= allow)) stats[1] += 1;
            }
            if (take) {
                member[i] = 1;
                W += in[i].weight;
                sx += in[i].weight * in[i].mean[0];
                sy += in[i].weight * in[i].mean[1];
                o_exact_first(&es, &in[i]);
            }
        }
        o_gaussian mg;
        if (cfg->mergeSums == 1) {
            if (W == 0) break;
            mg.weight = W; mg.mean[0] = sx / W; mg.mean[1] = sy / W;
        } else {
            if (es.ok && es.ec < 255 && es.W == 0) break;
            o_exact_mean(&es, &mg);
        }
        float c[4] = {0, 0, 0, 0};
        for (int k = first; k < n; k++) {
            const int i = order[k].idx;
            if (merged[i] || !member[i]) continue;
            const float d0 = mg.mean[0] - in[i].mean[0], d1 = mg.mean[1] - in[i].mean[1];
            const float dd[2] = {d0, d1};
            for (int j = 0; j < 2; j++)
                for (int kk = 0; kk < 2; kk++)
                    c[j * 2 + kk] += in[i].weight * (in[i].cov[j * 2 + kk] + dd[j] * dd[kk]);
            if (cfg->mergeSums != 1) o_exact_second(&es, &mg, &in[i]);
            merged[i] = 1;
        }
        if (cfg->mergeSums == 1) {
            for (int j = 0; j < 4; j++) mg.cov[j] = c[j] / W;
            mg.cov[1] = (mg.cov[1] + mg.cov[2]) / 2;
            mg.cov[2] = mg.cov[1];
        } else {
            o_exact_cov(&es, &mg);
        }
        out[n_out++] = mg;
        /* (a seed that is not close to itself under ref's decision stays unmerged and is picked again: the loop then ends
         *  with W == 0 exactly as o_merge's does, because the decision is ref's) */
    }
    o_tmp_leave(tmp_frame);
    return n_out;
}

/* ------------------------------------------------------------------------------------ */
/* literal transcription of reduceGaussianMixture, src/gm_reduce.cpp:57-134                */
/* (Eigen LLT distance :30-37 written out for 2x2; std::sort is not stable, a stable sort   */
/*  with index tie-break is used here)                                                     */
/* ------------------------------------------------------------------------------------ */
static float o_chol_dist(const o_gaussian* a, const o_gaussian* b)
{
    float d0 = a->mean[0] - b->mean[0], d1 = a->mean[1] - b->mean[1];
    /* cov(i,j) = g.cov[i+dims*j] (:24): s00=cov[0], s10=cov[1], s01=cov[2], s11=cov[3] */
    float s00 = 0.5f * (a->cov[0] + b->cov[0]);
    float s10 = 0.5f * (a->cov[1] + b->cov[1]);
    float s11 = 0.5f * (a->cov[3] + b->cov[3]);
    float l00 = sqrtf(s00);
    float l10 = s10 / l00;
    float l11 = sqrtf(s11 - l10 * l10);
    float x0 = d0 / l00;
    float x1 = (d1 - l10 * x0) / l11;
    return x0 * x0 + x1 * x1;
}

int o_gm_reduce(const o_gaussian* in, int n, float min_distance, o_gaussian* out)
{
    if (n <= 0) return 0;
    o_tmp_frame tmp_frame = o_tmp_enter();
    o_sortkey* order = (o_sortkey*)o_tmp_alloc(sizeof(o_sortkey) * n);
    uint8_t* gone = (uint8_t*)o_tmp_alloc(n);
    int* mlist = (int*)o_tmp_alloc(sizeof(int) * n);
    memset(gone, 0, n);
    for (int i = 0; i < n; i++) { order[i].w = in[i].weight; order[i].idx = i; }
    qsort(order, n, sizeof(o_sortkey), o_cmp_desc);                       /* :75-77 */
    int n_out = 0;
    for (int f = 0; f < n; f++) {
        if (gone[order[f].idx]) continue;
        const o_gaussian* mx = &in[order[f].idx];                        /* :81-82 */
        gone[order[f].idx] = 1;
        int nm = 0;
        for (int k = f + 1; k < n; k++) {                                /* :86-100 */
            int i = order[k].idx;
            if (gone[i]) continue;
            if (o_chol_dist(mx, &in[i]) < min_distance) { mlist[nm++] = i; gone[i] = 1; }
        }
        float W = mx->weight;                                            /* :103-109 */
        float m0 = mx->mean[0] * mx->weight, m1 = mx->mean[1] * mx->weight;
        for (int k = 0; k < nm; k++) {
            m0 += in[mlist[k]].weight * in[mlist[k]].mean[0];
            m1 += in[mlist[k]].weight * in[mlist[k]].mean[1];
            W += in[mlist[k]].weight;
        }
        m0 /= W; m1 /= W;
        float d0 = m0 - mx->mean[0], d1 = m1 - mx->mean[1];              /* :110-112 */
        float c00 = mx->weight * (mx->cov[0] + d0 * d0);
        float c10 = mx->weight * (mx->cov[1] + d1 * d0);
        float c01 = mx->weight * (mx->cov[2] + d0 * d1);
        float c11 = mx->weight * (mx->cov[3] + d1 * d1);
        for (int k = 0; k < nm; k++) {                                   /* :114-118 */
            const o_gaussian* g = &in[mlist[k]];
            d0 = m0 - g->mean[0]; d1 = m1 - g->mean[1];
            c00 += g->weight * (g->cov[0] + d0 * d0);
            c10 += g->weight * (g->cov[1] + d1 * d0);
            c01 += g->weight * (g->cov[2] + d0 * d1);
            c11 += g->weight * (g->cov[3] + d1 * d1);
        }
        o_gaussian r;
        r.weight = W;
        r.mean[0] = m0; r.mean[1] = m1;
        r.cov[0] = c00 / W; r.cov[1] = c10 / W; r.cov[2] = c01 / W; r.cov[3] = c11 / W; /* :119-129 */
        out[n_out++] = r;
    }
    o_tmp_leave(tmp_frame);
    return n_out;
}

/* ------------------------------------------------------------------------------------ */
/* one particle's measurement update, src/phdfilter.cu:3336-3761 restricted to one map     */
/* ------------------------------------------------------------------------------------ */
int o_update_particle(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                      const o_config* cfg, o_gaussian* map_out, float* dlogw,
                      o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out,
                      float* margin_out)
{
    return o_update_particle_ex(pose, map, n_map, z, M, cfg, map_out, dlogw, survivors_out, surv_slab_idx, n_survivors_out,
                                margin_out, NULL);
}

/* the same, and (test diagnostic) the whole UNPRUNED slab followed by the nearly-in-range features in slab_all_out
 * (n_in (M + 1) + M + n_near entries: what surv_slab_idx indexes) */
int o_update_particle_ex(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                         const o_config* cfg, o_gaussian* map_out, float* dlogw,
                         o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out,
                         float* margin_out, o_gaussian* slab_all_out)
{
    o_tmp_frame tmp_frame = o_tmp_enter();
    int8_t* cls = (int8_t*)o_tmp_alloc(n_map > 0 ? n_map : 1);
    o_classify(map, n_map, pose, cfg, cls);
    int n_in = 0, n_near = 0, n_out0 = 0;
    for (int i = 0; i < n_map; i++) { n_in += cls[i] == 1; n_near += cls[i] == 2; n_out0 += cls[i] == 0; }
    /* stable 3-way partition, src/phdfilter.cu:3048-3056 */
    o_gaussian* f_in = (o_gaussian*)o_tmp_alloc(sizeof(o_gaussian) * (n_in + 1));
    o_gaussian* f_near = (o_gaussian*)o_tmp_alloc(sizeof(o_gaussian) * (n_near + 1));
    o_gaussian* f_out = (o_gaussian*)o_tmp_alloc(sizeof(o_gaussian) * (n_out0 + 1));
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < n_map; i++) {
        if (cls[i] == 1) f_in[a++] = map[i];
        else if (cls[i] == 2) f_near[b++] = map[i];
        else f_out[c++] = map[i];
    }
    size_t n_update = (size_t)n_in * (M + 1) + M;
    o_gaussian* births = (o_gaussian*)o_tmp_alloc(sizeof(o_gaussian) * (M + 1));
    o_gaussian* pre = (o_gaussian*)o_tmp_alloc(sizeof(o_gaussian) * ((size_t)n_in * M + 1));
    float* pd = (float*)o_tmp_alloc(sizeof(float) * (n_in + 1));
    o_gaussian* slab = (o_gaussian*)o_tmp_alloc(sizeof(o_gaussian) * (n_update + n_near + 1));
    uint8_t* flag = (uint8_t*)o_tmp_alloc(n_update + 1);
    o_births(pose, z, M, cfg, births);
    o_preupdate(pose, f_in, n_in, z, M, cfg, pd, pre);
    o_update(f_in, pd, pre, births, n_in, M, cfg, slab, flag, dlogw);
    if (slab_all_out) {
        memcpy(slab_all_out, slab, sizeof(o_gaussian) * n_update);
        memcpy(slab_all_out + n_update, f_near, sizeof(o_gaussian) * n_near);
    }
    /* prune: stable compaction (thrust::remove_copy_if, src/phdfilter.cu:3134-3137) */
    int ns = 0;
    for (size_t i = 0; i < n_update; i++) {
        if (!flag[i]) {
            if (surv_slab_idx) surv_slab_idx[ns] = (int32_t)i;
            slab[ns++] = slab[i];
        }
    }
    /* recombine with the nearly-in-range features (src/phdfilter.cu:3227-3257) */
    for (int i = 0; i < n_near; i++) {
        if (surv_slab_idx) surv_slab_idx[ns] = (int32_t)(n_update + i);
        slab[ns++] = f_near[i];
    }
    if (survivors_out) memcpy(survivors_out, slab, sizeof(o_gaussian) * ns);
    if (n_survivors_out) *n_survivors_out = ns;
    int nm = o_merge(slab, ns, cfg, map_out, margin_out);
    /* append the out-of-range features (src/phdfilter.cu:3311-3318) */
    for (int i = 0; i < n_out0; i++) map_out[nm++] = f_out[i];
    o_tmp_leave(tmp_frame);
    return nm;
}

/* ------------------------------------------------------------------------------------ */
/* particle weights                                                                       */
/* ------------------------------------------------------------------------------------ */
void o_normalize_weights(float* logw, const float* dlogw, int n)
{
    if (dlogw) for (int i = 0; i < n; i++) logw[i] += dlogw[i];          /* src/phdfilter.cu:3741-3744 */
    float maxval = logw[0];                                              /* logSumExp, device_math.cuh:549-558 */
    for (int i = 1; i < n; i++) if (logw[i] > maxval) maxval = logw[i];
    float sum = 0;
    for (int i = 0; i < n; i++) sum += expf(logw[i] - maxval);
    float lse = o_safe_log(sum) + maxval;
    for (int i = 0; i < n; i++) logw[i] -= lse;                          /* :3751-3754 */
}

float o_neff(const float* logw, int n)
{
    float nEff = 0;                                                      /* src/main.cpp:1281-1284 */
    for (int i = 0; i < n; i++) nEff += expf(2 * logw[i]);
    nEff = (float)(1.0 / (double)nEff / (double)n);
    return nEff;
}

/*
 * Resampling (src/main.cpp:453-501; systematic: src/phdfilter.cu.bak:3279-3327).
 *
 * The reference walks a CDF accumulated in double, "c += exp(w_i)" (:463,495), against thresholds
 * r_j = j*interval + u_j*interval (:468).  A floating-point running sum depends on the order of
 * the additions, so no parallel implementation can reproduce it bit for bit, and its libm exp()
 * is platform specific.  The CDF here is the same quantity in FIXED POINT:
 *     sb  = 62 - ceil(log2 N)                       (so that the total fits 63 bits)
 *     q_i = floor(min(det_exp(w_i), 1) * 2^sb)      (exact: power-of-two scaling, then floor)
 *     Q_i = q_0 + ... + q_i                         (exact integer sum: associative)
 *     "r_j > c_i"  <=>  Q_i < T_j,  T_j = ceil(r_j * 2^sb)
 * Resolution N*2^-62 of the total mass, a few hundred times finer than the rounding noise of the
 * reference's own double sum (N*2^-53).  Every implementation — sequential, parallel scan, any
 * rank — gets the same indices.  Weights above exp(0) are clamped (normalised weights never are).
 */
static int o_ceil_log2(int n) { int b = 0; while ((1LL << b) < n) b++; return b; }

void o_resample(const float* logw, int n, const double* uniforms, int n_uniforms, int n_new, int32_t* idx)
{
    const int sb = 62 - o_ceil_log2(n);
    const double scale = ldexp(1.0, sb);
    double interval = 1.0 / n_new;                                       /* src/main.cpp:461 */
    double p0 = o_det_exp(logw[0]);
    uint64_t c = (uint64_t)floor((p0 > 1.0 ? 1.0 : p0) * scale);          /* :463 */
    int i = 0;
    int overflowed = 0;
    for (int j = 0; j < n_new; j++) {
        /* :468 "r = j*interval + randu01()*interval": one uniform per stratum (stratified, HEAD) or the
         * same uniform for every j (n_uniforms == 1: systematic) */
        double r = j * interval + uniforms[n_uniforms == 1 ? 0 : j] * interval;
        uint64_t T = (uint64_t)ceil(r * scale);
        while (!overflowed && c < T) {                                   /* :469  r > c */
            i++;
            if (i >= n || i < 0) {                                       /* :475-490 */
                double max_weight = -1;
                int max_idx = -1;
                for (int k = 0; k < n; k++) {
                    double e = o_det_exp(logw[k]);
                    if (e > max_weight) { max_weight = e; max_idx = k; }
                }
                i = max_idx;
                overflowed = 1;                                          /* ":492 c = 2": never enter again */
                break;
            }
            double p = o_det_exp(logw[i]);
            c += (uint64_t)floor((p > 1.0 ? 1.0 : p) * scale);            /* :495 */
        }
        idx[j] = i;
    }
}

void o_expected_pose(const o_pose* poses, const float* logw, int n, o_pose* out)
{
    if (n == 1) { *out = poses[0]; return; }                             /* src/main.cpp:381-384 */
    o_pose e = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {                                        /* :331-340 */
        float w = expf(logw[i]);
        e.px += w * poses[i].px;
        e.py += w * poses[i].py;
        e.ptheta += w * poses[i].ptheta;
        e.vx += w * poses[i].vx;
        e.vy += w * poses[i].vy;
        e.vtheta += w * poses[i].vtheta;
    }
    *out = e;
}

int o_argmax_weight(const float* logw, int n)
{
    float max_weight = -FLT_MAX;                                          /* src/main.cpp:347-356 */
    int max_idx = -1;
    for (int i = 0; i < n; i++)
        if (logw[i] > max_weight) { max_idx = i; max_weight = logw[i]; }
    return max_idx;
}

/* computeExpectedMap, src/main.cpp:290-316: concatenate the particle maps (sizes[n] Gaussians each,
 * back to back in `maps`) with weights scaled by exp(particle log-weight), then reduceGaussianMixture.
 * exp is the portable o_det_exp rounded to float (the reference's libm expf is not reproducible
 * across machines; the two agree to 1 ulp). */
int o_expected_map(const o_gaussian* maps, const int32_t* sizes, const float* logw, int n_particles,
                   float min_distance, o_gaussian* out)
{
    long total = 0;
    for (int n = 0; n < n_particles; n++) total += sizes[n];
    if (total == 0) return 0;                                             /* :308-313 */
    o_gaussian* concat = (o_gaussian*)malloc(sizeof(o_gaussian) * (size_t)total);
    long t = 0;
    for (int n = 0; n < n_particles; n++) {
        const float f = (float)o_det_exp(logw[n]);
        for (int i = 0; i < sizes[n]; i++, t++) {
            concat[t] = maps[t];
            concat[t].weight = maps[t].weight * f;                        /* :303 */
        }
    }
    int k = o_gm_reduce(concat, (int)total, min_distance, out);           /* :315 */
    free(concat);
    return k;
}

/* ------------------------------------------------------------------------------------ */
/* whole step (run_synth loop body, src/main.cpp:1244-1297) on fixed-capacity slabs        */
/* ------------------------------------------------------------------------------------ */
int o_step(o_pose* poses, float* logw, o_gaussian* maps, int32_t* sizes, int n_particles, int cap,
           float alpha, float v_encoder, const float* noise, const o_meas* z, int M,
           const o_config* cfg, double uniform, int force_resample,
           o_gaussian* maps_out, int32_t* sizes_out, int32_t* idx_out, float* neff_out, int n_threads)
{
    int overflow = 0;
    o_predict_ackerman(poses, n_particles, alpha, v_encoder, noise, cfg);
    float* dlogw = (float*)malloc(sizeof(float) * n_particles);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        size_t tmp_n = (size_t)cap * (M + 2) + M + 1;
        o_gaussian* tmp = (o_gaussian*)malloc(sizeof(o_gaussian) * tmp_n);
#pragma omp for schedule(dynamic, 4)
        for (int p = 0; p < n_particles; p++) {
            int nm = o_update_particle(&poses[p], maps + (size_t)p * cap, sizes[p], z, M, cfg, tmp, &dlogw[p],
                                       NULL, NULL, NULL, NULL);
            if (nm > cap) {
#pragma omp atomic write
                overflow = 1;
                nm = cap;
            }
            memcpy(maps_out + (size_t)p * cap, tmp, sizeof(o_gaussian) * nm);
            sizes_out[p] = nm;
        }
        free(tmp);
    }
    o_normalize_weights(logw, dlogw, n_particles);
    float neff = o_neff(logw, n_particles);
    if (neff_out) *neff_out = neff;
    if (force_resample || (neff <= cfg->resampleThresh && M > 0)) {
        o_resample(logw, n_particles, &uniform, 1, n_particles, idx_out);
    } else {
        for (int i = 0; i < n_particles; i++) idx_out[i] = i;
    }
    free(dlogw);
    return overflow ? -1 : 0;
}
