/*
 * cphd_cpu.c — CPU ORACLE of the CPHD variant (test infrastructure, not product code).
 *
 * PARITY UNPINNED.  The reference's HEAD has no runnable CPHD: the kernels are commented out in
 * src/phdfilter.cu (:701-779 constants, :1360-1591 cardinality/ESF) and filter_type is never read
 * by phdUpdateSynth.  The only complete statement is the older src/phdfilter.cu.bak
 * (cphdConstantsKernel :369-415, cardinalityPredictKernel :518-545, birth cardinality :779-790,
 * cphdPreUpdateKernel :1058-1176, computeEsfKernel :1191-1274, computePsiKernel :1282-1412,
 * cphdUpdateKernel :1420-1462, host sequence :2388-2544,2661-2709).  That version has defects
 * (leave-one-out ESFs subtract magnitudes :1259; the psi1d inner product mixes two maxima
 * :1401-1404; the predicted cardinality is overwritten by a Poisson law :2470-2484, which makes
 * the update algebraically the PHD update) and a different birth scheme from HEAD.  This file
 * therefore states the algorithm those kernels implement — the Gaussian-mixture CPHD recursion of
 * Vo, Vo & Cantoni (IEEE TSP 55(7), 2007, eqs. 31-35, 44-49) — in the .bak's decomposition and
 * log-domain arithmetic, on HEAD's update structure:
 *
 *   scope           the particle's whole map + the M measurement-driven births of HEAD
 *                   (src/phdfilter.cu:3465-3510); p_D = pd for in-range features, 0 for the rest,
 *                   1 for a birth with respect to its own measurement
 *   cardinality     log p(n), n = 0..maxCardinality, per particle (.bak:1142 uniform start);
 *                   predicted = prior (*) Binomial(M, birthWeight)      (.bak:518-545, 779-790)
 *   roots           Xi_m = (clutterRate/clutterDensity) (sum_j pd w_j g_jm + birthWeight)  (.bak:1205-1222)
 *   ESFs            e_j(Xi), e_j(Xi \ m) by the log-domain recursion                       (.bak:1224-1272)
 *   Upsilon^u(n)    sum_j (M-j)! p_K(M-j) P(n,j+u) <1-pD,v>^(n-j-u) / <1,v>^n e_j          (.bak:1342-1366)
 *   weights         missed: w (1-pD) <Y1,p>/<Y0,p>;  detected: pd w g (lambda/kappa) <Y1[Z\m],p>/<Y0,p>
 *                                                                                          (.bak:1420-1462)
 *   cardinality     p(n) Y0(n) / <Y0,p>                                                     (.bak:1409-1411)
 *   particle weight += log <Y0,p>                                                           (.bak:2661-2667)
 * With a Poisson prior of mean <1,v> this reduces to HEAD's PHD update exactly (tested).
 * Checked against an independent float64 brute-force evaluation (subset enumeration) in tests/.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "scphd_cpu.h"

#define LOG0 (-FLT_MAX)

static float clampl(float x) { return x < -1e30f ? -1e30f : x; }

/* 1: leave-one-out ESFs by M separate recursions (the .bak's O(M^3) structure) instead of the O(M^2)
 * prefix/suffix form — kept as an independent check of the latter (tests/test_cphd_oracle.py) */
static int g_reference_esf = 0;
void o_cphd_set_reference_esf(int on) { g_reference_esf = on; }

/* A positive number spanning hundreds of decades: float mantissa, separate integer exponent (m 2^k).
 * Additions align with ldexpf and renormalise with frexpf — exact operations around one correctly
 * rounded multiply and add, hence identical on every IEEE machine (the device uses v_ldexp / v_frexp). */
typedef struct { float m; int k; } xf;
#define XF_ZERO_K (-(1 << 28))

static xf xf_axpy(xf a, xf b, float x)                              /* a + x b */
{
    const float tm = b.m * x;
    const int k = a.k > b.k ? a.k : b.k;
    const float s = ldexpf(a.m, a.k - k) + ldexpf(tm, b.k - k);
    int dk = 0;
    xf r;
    r.m = frexpf(s, &dk);
    r.k = k + dk;
    return r;
}

static float xf_log(xf a) { return a.m > 0 ? logf(a.m) + (float)a.k * 0.69314718f : LOG0; }

static xf xf_from_log(float L)
{
    xf r;
    if (!(L > -1e30f)) { r.m = 0; r.k = XF_ZERO_K; return r; }
    const double t = (double)L * 1.4426950408889634;
    const double kf = ceil(t);
    r.m = (float)exp2(t - kf);                                       /* in (0.5, 1] */
    r.k = (int)kf;
    return r;
}

/* log sum exp of t[0..n) (two passes: max, then the sum in index order) */
static float lse_n(const float* t, int n)
{
    if (n <= 0) return LOG0;
    float mx = t[0];
    for (int i = 1; i < n; i++) if (t[i] > mx) mx = t[i];
    float s = 0;
    for (int i = 0; i < n; i++) s += expf(t[i] - mx);
    return o_safe_log(s) + mx;
}

void o_cphd_log_factorials(float* lfact, int n)                      /* initCphdConstants, .bak:421-425 */
{
    lfact[0] = 0;
    for (int k = 1; k < n; k++) lfact[k] = lfact[k - 1] + o_safe_log((float)k);
}

/* ESF of the roots xi[0..M) without root `skip` (skip < 0: none): e_j <- e_j + xi_m e_{j-1}, the
 * recursion of computeEsfKernel (.bak:1224-1272; linear domain as in HEAD's commented version,
 * src/phdfilter.cu:1555-1579).  e_j spans hundreds of decades, so each value is a float mantissa with
 * a separate integer exponent (m * 2^k, m in [0.5,1) or 0); additions align with ldexpf and renormalise
 * with frexpf — exact operations around one correctly rounded add and multiply, hence identical on
 * every IEEE machine.  Output: log e_j, j = 0..M. */
static void esf_xf(const float* xi, int M, int skip, float* le)
{
    o_tmp_frame tmp_frame = o_tmp_enter();
    xf* e = (xf*)o_tmp_alloc(sizeof(xf) * (M + 1));
    e[0].m = 0.5f; e[0].k = 1;                                       /* e_0 = 1 */
    for (int j = 1; j <= M; j++) { e[j].m = 0; e[j].k = XF_ZERO_K; }
    int done = 0;
    for (int m = 0; m < M; m++) {
        if (m == skip) continue;
        for (int j = done + 1; j >= 1; j--) e[j] = xf_axpy(e[j], e[j - 1], xi[m]);
        done++;
    }
    for (int j = 0; j <= M; j++) le[j] = xf_log(e[j]);
    o_tmp_leave(tmp_frame);
}

/* ------------------------------------------------------------------------------------ */
/* The same recursions in DOUBLE with a block exponent per row (scans of up to 64 measurements; round 3).
 * A step is one fused multiply-add per entry, e_j <- fma(xi_m, e_{j-1}, e_j); every sixteenth step the row is rescaled by a
 * power of two so that its largest entry sits at 2^0 (exact; entries 2^-1074 below the largest flush to zero), the shift
 * accumulated in an integer.  The device (cuda-phdslam_amd/csrc/phd_cphd.h, cphd_esf_*_f64) runs the same steps, the same
 * rescaling schedule and the same fma; only the order of the 64-term sums of the inner products differs (a wave's
 * butterfly there, index order here), i.e. agreement to ~1e-15 before the logarithm is rounded to float. */
/* ------------------------------------------------------------------------------------ */
#define F64_RENORM 16

static int dexp_field(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)((b >> 52) & 0x7FF); }

static float log_scaled(double v, int k)
{
    if (!(v > 0.0)) return LOG0;
    int e = 0;
    const double mant = frexp(v, &e);
    return logf((float)mant) + (float)(e + k) * 0.69314718f;
}

/* rescale row[0..n) so that its largest exponent becomes 0 (biased 1023); returns the shift taken out */
static int row_renorm(double* row, int n)
{
    int r = 0;
    for (int a = 0; a < n; a++) { const int f = dexp_field(row[a]); if (f > r) r = f; }
    r -= 1023;
    if (r <= -1023) return 0;                                        /* an all-zero row */
    for (int a = 0; a < n; a++) row[a] = ldexp(row[a], -r);
    return r;
}

/* log e_j(xi[0..M)), j = 0..M */
static void esf_f64(const float* xi, int M, float* le)
{
    double P[66];
    int kP = 0;
    P[0] = 1.0;
    for (int j = 1; j <= M; j++) P[j] = 0.0;
    for (int m = 0; m < M; m++) {
        for (int j = m + 1; j >= 1; j--) P[j] = fma((double)xi[m], P[j - 1], P[j]);
        if ((m & (F64_RENORM - 1)) == F64_RENORM - 1) kP += row_renorm(P, M + 1);
    }
    le[0] = 0.f;
    for (int j = 1; j <= M; j++) le[j] = log_scaled(P[j], kP);
}

/* lz[m] = -((llam - lkap) + log <Y1[Z\m],p> - lY0) for every m, prefix/suffix form in double */
static void leave_one_out_f64(const float* xi, const float* I1, int M, float llam, float lam, float lkap, float lY0, float* lz)
{
    o_tmp_frame tmp_frame = o_tmp_enter();
    double* rows = (double*)o_tmp_alloc(sizeof(double) * (size_t)M * 65);
    int kp[64];
    double P[66];
    int kP = 0;
    P[0] = 1.0;
    for (int j = 1; j <= M; j++) P[j] = 0.0;
    for (int m = 0; m < M; m++) {                                    /* forward: park P_m[0..m] */
        for (int a = 0; a <= m; a++) rows[(size_t)m * 65 + a] = P[a];
        kp[m] = kP;
        for (int j = m + 1; j >= 1; j--) P[j] = fma((double)xi[m], P[j - 1], P[j]);
        if ((m & (F64_RENORM - 1)) == F64_RENORM - 1) kP += row_renorm(P, M + 1);
    }
    /* T_M[a] = c_a = exp(I1[a]) lambda^(M-1-a) e^-lambda on a common block exponent */
    double T[65], frac[64];
    int kf[64], kT = -(1 << 28);
    for (int a = 0; a < M; a++) {
        const float Lg = I1[a] + ((float)(M - 1 - a) * llam - lam);
        kf[a] = -(1 << 28); frac[a] = 0.0;
        if (Lg > -1e30f) {
            const double t = (double)Lg * 1.4426950408889634;
            const double kc = ceil(t);
            frac[a] = exp2(t - kc);
            kf[a] = (int)kc;
        }
        if (kf[a] > kT) kT = kf[a];
    }
    if (kT == -(1 << 28)) kT = 0;
    for (int a = 0; a < M; a++) {
        T[a] = 0.0;
        if (kf[a] != -(1 << 28)) { const int d = kf[a] - kT; T[a] = d < -1100 ? 0.0 : ldexp(frac[a], d); }
    }
    T[M] = 0.0;
    for (int m = M - 1; m >= 0; m--) {
        double s = 0.0;
        for (int a = 0; a <= m; a++) s += rows[(size_t)m * 65 + a] * T[a];
        lz[m] = -((llam - lkap) + log_scaled(s, kp[m] + kT) - lY0);
        if (m >= 1) {
            for (int a = 0; a <= m - 1; a++) T[a] = fma((double)xi[m], T[a + 1], T[a]);
            if (((M - 1 - m) & (F64_RENORM - 1)) == F64_RENORM - 1) kT += row_renorm(T, M);
        }
    }
    o_tmp_leave(tmp_frame);
}

/*
 * The cardinality-dependent terms of one particle's CPHD update.
 *   cn_prior[cn_len]  log cardinality before the births of this step
 *   S[M]              sum_j exp(lw_jm) (linear), the detection masses of each measurement
 *   w_all, pdw        <1,map>, <pD,map>
 * Outputs: lz[M] (detection weight = exp(lw_jm - lz[m])), r1 (missed-detection factor),
 * cn_out[cn_len], lY0 (= particle log-weight increment).
 */
void o_cphd_terms(const float* cn_prior, int cn_len, const float* S, int M, float w_all, float pdw,
                  float birth_weight, float clutter_rate, float clutter_density,
                  float* lz, float* r1_out, float* cn_out, float* lY0_out)
{
    const int Nmax = cn_len - 1;
    const int LF = (Nmax > M ? Nmax : M) + 2;
    o_tmp_frame tmp_frame = o_tmp_enter();
    float* lfact = (float*)o_tmp_alloc(sizeof(float) * LF);
    float* cnp = (float*)o_tmp_alloc(sizeof(float) * cn_len);
    float* cnb = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* lxi = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* e = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* em = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* I0 = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* I1 = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* t = (float*)o_tmp_alloc(sizeof(float) * (cn_len + M + 2));
    o_cphd_log_factorials(lfact, LF);
    const float llam = o_safe_log(clutter_rate), lkap = o_safe_log(clutter_density);
    const float lbw = o_safe_log(birth_weight), l1bw = o_safe_log(1 - birth_weight);
    const float W1 = w_all + (float)M * birth_weight;               /* <1,v>: map + births */
    const float Wq = w_all - pdw;                                   /* <1-pD,v>: births are always detected */
    const float lW1 = clampl(o_safe_log(W1)), lWq = clampl(o_safe_log(Wq));

    /* 1. predicted cardinality = prior (*) Binomial(M, birthWeight)  (.bak:518-545 with :779-790) */
    const int Kb = M < Nmax ? M : Nmax;
    for (int k = 0; k <= Kb; k++)
        cnb[k] = lfact[M] - lfact[k] - lfact[M - k] + (float)k * lbw + (float)(M - k) * l1bw;
    for (int n = 0; n <= Nmax; n++) {
        const int kmax = n < Kb ? n : Kb;
        for (int k = 0; k <= kmax; k++) t[k] = cnb[k] + cn_prior[n - k];
        cnp[n] = lse_n(t, kmax + 1);
    }
    /* 2. roots Xi_m = (lambda/kappa)(S_m + w_b) (.bak:1205-1222) */
    const float rat = clutter_rate / clutter_density;
    for (int m = 0; m < M; m++) lxi[m] = (S[m] + birth_weight) * rat;
    /* 3. inner products over n that do not depend on the measurements:
     *    I_u[j] = log sum_n p(n) P(n, j+u) Wq^(n-j-u) / W1^n */
    for (int j = 0; j <= M; j++) {
        int c = 0;
        for (int n = j; n <= Nmax; n++)
            t[c++] = cnp[n] + (lfact[n] - lfact[n - j]) + (float)(n - j) * lWq - (float)n * lW1;
        I0[j] = lse_n(t, c);
        c = 0;
        for (int n = j + 1; n <= Nmax; n++)
            t[c++] = cnp[n] + (lfact[n] - lfact[n - j - 1]) + (float)(n - j - 1) * lWq - (float)n * lW1;
        I1[j] = lse_n(t, c);
    }
    /* 4. full ESF, <Y0,p>, <Y1,p>;  (M-j)! p_K(M-j) = lambda^(M-j) e^-lambda for Poisson clutter (.bak:398-400) */
    if (M <= 64 && !g_reference_esf) esf_f64(lxi, M, e);
    else esf_xf(lxi, M, -1, e);
    for (int j = 0; j <= M; j++) t[j] = e[j] + I0[j] + ((float)(M - j) * llam - clutter_rate);
    const float lY0 = lse_n(t, M + 1);
    for (int j = 0; j <= M; j++) t[j] = e[j] + I1[j] + ((float)(M - j) * llam - clutter_rate);
    const float lY1 = lse_n(t, M + 1);
    /* 5. <Y1[Z\m],p> = sum_j e_j(Xi \ m) c_j,  c_j = exp(I1[j]) lambda^(M-1-j) e^-lambda. */
    if (g_reference_esf) {
        /* the .bak's way (:1247-1272): one full ESF recursion per left-out measurement, O(M^3) */
        for (int m = 0; m < M; m++) {
            esf_xf(lxi, M, m, em);
            for (int j = 0; j <= M - 1; j++) t[j] = em[j] + I1[j] + ((float)(M - 1 - j) * llam - clutter_rate);
            const float lD = lse_n(t, M);
            lz[m] = -((llam - lkap) + lD - lY0);
        }
    } else if (M <= 64) {
        /* the O(M^2) prefix/suffix form below, in double with block exponents (what the device runs for M <= 64) */
        leave_one_out_f64(lxi, I1, M, llam, clutter_rate, lkap, lY0, lz);
    } else {
        /* O(M^2): e(Xi \ m) = P_m (*) S_{m+1} (ESFs of the roots before and after m), hence
         *   <Y1[Z\m],p> = sum_a P_m[a] T_{m+1}[a],   T_{m+1}[a] = sum_b S_{m+1}[b] c_{a+b},
         * and T obeys the same one-root recursion run backwards: T_m[a] = T_{m+1}[a] + xi_m T_{m+1}[a+1],
         * T_M = c.  All terms are positive: no cancellation.  (The device kernel does exactly this.) */
        xf* T = (xf*)o_tmp_alloc(sizeof(xf) * (size_t)M * M);             /* row m holds T_{m+1}[0..m] */
        xf* cur = (xf*)o_tmp_alloc(sizeof(xf) * (M + 1));
        xf* P = (xf*)o_tmp_alloc(sizeof(xf) * (M + 1));
        for (int a = 0; a < M; a++) cur[a] = xf_from_log(I1[a] + ((float)(M - 1 - a) * llam - clutter_rate));
        for (int a = 0; a < M; a++) T[(size_t)(M - 1) * M + a] = cur[a];
        for (int m = M - 1; m >= 1; m--) {
            for (int a = 0; a <= m - 1; a++) cur[a] = xf_axpy(cur[a], cur[a + 1], lxi[m]);
            for (int a = 0; a <= m - 1; a++) T[(size_t)(m - 1) * M + a] = cur[a];
        }
        P[0].m = 0.5f; P[0].k = 1;                                    /* P_0 = [1] */
        for (int a = 1; a <= M; a++) { P[a].m = 0; P[a].k = XF_ZERO_K; }
        for (int m = 0; m < M; m++) {
            /* dot product in the mantissa+exponent form: align to the largest exponent, add, renormalise */
            int K = XF_ZERO_K * 2;
            for (int a = 0; a <= m; a++) { const int k = P[a].k + T[(size_t)m * M + a].k; if (k > K) K = k; }
            float s = 0;
            for (int a = 0; a <= m; a++)
                s += ldexpf(P[a].m * T[(size_t)m * M + a].m, P[a].k + T[(size_t)m * M + a].k - K);
            int dk = 0;
            xf D; D.m = frexpf(s, &dk); D.k = K + dk;
            lz[m] = -((llam - lkap) + xf_log(D) - lY0);
            for (int a = m + 1; a >= 1; a--) P[a] = xf_axpy(P[a], P[a - 1], lxi[m]);
        }
        /* (T, cur, P: released with the frame) */
    }
    *r1_out = expf(lY1 - lY0);
    /* 6. updated cardinality (.bak:1409-1411) */
    for (int n = 0; n <= Nmax; n++) {
        const int jmax = n < M ? n : M;
        for (int j = 0; j <= jmax; j++)
            t[j] = e[j] + ((float)(M - j) * llam - clutter_rate) + (lfact[n] - lfact[n - j]) + (float)(n - j) * lWq
                   - (float)n * lW1;
        cn_out[n] = cnp[n] + lse_n(t, jmax + 1) - lY0;
    }
    *lY0_out = lY0;
    o_tmp_leave(tmp_frame);
}

/* o_update for the CPHD variant: same slab layout [non-detect | detect m-major | births] */
void o_cphd_update(const o_gaussian* feat, const float* pd, const o_gaussian* preupdate, const o_gaussian* births,
                   int n, int M, const o_config* cfg, float clutter_rate, float w_all,
                   const float* cn_prior, int cn_len,
                   o_gaussian* slab, uint8_t* prune_flag, float* dlogw, float* cn_out, float* r1_out)
{
    const int n_update = n * (M + 1) + M;
    o_tmp_frame tmp_frame = o_tmp_enter();
    float* S = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float* lz = (float*)o_tmp_alloc(sizeof(float) * (M + 1));
    float pdw = 0;
    for (int j = 0; j < n; j++) pdw += pd[j] * feat[j].weight;
    for (int m = 0; m < M; m++) {
        float sum = 0;
        for (int j = 0; j < n; j++) sum += expf(preupdate[(size_t)m * n + j].weight);
        S[m] = sum;
    }
    float r1 = 1, lY0 = 0;
    o_cphd_terms(cn_prior, cn_len, S, M, w_all, pdw, cfg->birthWeight, clutter_rate, cfg->clutterDensity,
                 lz, &r1, cn_out, &lY0);
    for (int j = 0; j < n; j++) {
        slab[j] = feat[j];
        slab[j].weight = feat[j].weight * (1 - pd[j]) * r1;
        for (int m = 0; m < M; m++) {
            o_gaussian g = preupdate[(size_t)m * n + j];
            g.weight = expf(g.weight - lz[m]);
            slab[n + (size_t)m * n + j] = g;
        }
    }
    for (int m = 0; m < M; m++) {
        o_gaussian b = births[m];
        b.weight = expf(b.weight - lz[m]);
        slab[n + (size_t)M * n + m] = b;
    }
    *dlogw = lY0;
    if (r1_out) *r1_out = r1;
    for (int i = 0; i < n_update; i++) prune_flag[i] = (slab[i].weight < cfg->minFeatureWeight) ? 1 : 0;
    o_tmp_leave(tmp_frame);
}

/* o_update_particle for the CPHD variant; features outside the field of view (pD = 0) take the
 * missed-detection factor r1 like every undetected part of the intensity */
int o_cphd_update_particle(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                           const o_config* cfg, float clutter_rate, const float* cn_prior, int cn_len,
                           o_gaussian* map_out, float* dlogw, float* cn_out,
                           o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out, float* r1_out)
{
    return o_cphd_update_particle_ex(pose, map, n_map, z, M, cfg, clutter_rate, cn_prior, cn_len, map_out, dlogw, cn_out,
                                     survivors_out, surv_slab_idx, n_survivors_out, r1_out, NULL, NULL);
}

/* the same, and (test diagnostics) margin_out[3] = o_merge's two decision margins + the smallest relative distance of an
 * update component's weight to min_feature_weight; slab_all_out = the whole UNPRUNED slab followed by the nearly-in-range
 * features (what surv_slab_idx indexes) */
int o_cphd_update_particle_ex(const o_pose* pose, const o_gaussian* map, int n_map, const o_meas* z, int M,
                              const o_config* cfg, float clutter_rate, const float* cn_prior, int cn_len,
                              o_gaussian* map_out, float* dlogw, float* cn_out,
                              o_gaussian* survivors_out, int32_t* surv_slab_idx, int* n_survivors_out, float* r1_out,
                              float* margin_out, o_gaussian* slab_all_out)
{
    o_tmp_frame tmp_frame = o_tmp_enter();
    int8_t* cls = (int8_t*)o_tmp_alloc(n_map > 0 ? n_map : 1);
    o_classify(map, n_map, pose, cfg, cls);
    int n_in = 0, n_near = 0, n_out0 = 0;
    float w_all = 0;
    for (int i = 0; i < n_map; i++) {
        n_in += cls[i] == 1; n_near += cls[i] == 2; n_out0 += cls[i] == 0;
        w_all += map[i].weight;
    }
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
    float r1 = 1;
    o_births(pose, z, M, cfg, births);
    o_preupdate(pose, f_in, n_in, z, M, cfg, pd, pre);
    o_cphd_update(f_in, pd, pre, births, n_in, M, cfg, clutter_rate, w_all, cn_prior, cn_len, slab, flag, dlogw, cn_out, &r1);
    if (slab_all_out) memcpy(slab_all_out, slab, sizeof(o_gaussian) * n_update);
    if (margin_out) {
        float pm = FLT_MAX;
        if (cfg->minFeatureWeight > 0)
            for (size_t i = 0; i < n_update; i++) {
                const float m = fabsf(slab[i].weight - cfg->minFeatureWeight) / cfg->minFeatureWeight;
                if (m < pm) pm = m;
            }
        margin_out[2] = pm;
    }
    int ns = 0;
    for (size_t i = 0; i < n_update; i++) {
        if (!flag[i]) {
            if (surv_slab_idx) surv_slab_idx[ns] = (int32_t)i;
            slab[ns++] = slab[i];
        }
    }
    for (int i = 0; i < n_near; i++) {
        o_gaussian g = f_near[i];
        g.weight = g.weight * r1;                                    /* joins the merge unpruned, like HEAD (:3242-3257) */
        if (surv_slab_idx) surv_slab_idx[ns] = (int32_t)(n_update + i);
        if (slab_all_out) slab_all_out[n_update + i] = g;
        slab[ns++] = g;
    }
    if (survivors_out) memcpy(survivors_out, slab, sizeof(o_gaussian) * ns);
    if (n_survivors_out) *n_survivors_out = ns;
    int nm = o_merge(slab, ns, cfg, map_out, margin_out);
    for (int i = 0; i < n_out0; i++) {
        map_out[nm] = f_out[i];
        map_out[nm].weight = f_out[i].weight * r1;
        nm++;
    }
    if (r1_out) *r1_out = r1;
    o_tmp_leave(tmp_frame);
    return nm;
}

/* whole CPHD filter step over fixed-capacity slabs, OpenMP over particles — o_step with the CPHD
 * update and the per-particle cardinality rows cn[p*cn_len ..) (the timed CPU baseline of config 5);
 * the rows follow the resampled particles (copy_particles, src/slamtypes.h:313-333) */
#ifdef _OPENMP
#include <omp.h>
#endif
int o_cphd_step(o_pose* poses, float* logw, o_gaussian* maps, int32_t* sizes, int n_particles, int cap,
                float alpha, float v_encoder, const float* noise, const o_meas* z, int M,
                const o_config* cfg, float clutter_rate, const float* cn, int cn_len, double uniform, int force_resample,
                o_gaussian* maps_out, int32_t* sizes_out, float* cn_out, int32_t* idx_out, float* neff_out, int n_threads)
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
            int nm = o_cphd_update_particle(&poses[p], maps + (size_t)p * cap, sizes[p], z, M, cfg, clutter_rate,
                                            cn + (size_t)p * cn_len, cn_len, tmp, &dlogw[p], cn_out + (size_t)p * cn_len,
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
