// phd_detexp.h — portable exp shared by the weight/resample kernels (phd_kernels.hip) and the
// expected-map concatenation (phd_eap.hip).
#ifndef PHD_DETEXP_H
#define PHD_DETEXP_H
#include <hip/hip_runtime.h>
#include <math.h>

namespace phd {
// ------------------------------------------------------------------------------------------
// portable exp for the resampling CDF: IEEE basic operations only (mul, fma, rint, ldexp), so
// the double it returns is the same on every conforming CPU and GPU (see oracle/scphd_cpu.c).
__device__ __forceinline__ double det_exp(float xf)
{
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double x = (double)xf;
    if (!(x >= -700.0)) return (x != x) ? x : 0.0;
    if (x > 700.0) return (double)INFINITY;
    double kd = rint(x * LOG2E);
    double r = fma(-kd, LN2_HI, x);
    r = fma(-kd, LN2_LO, r);
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

} // namespace phd
#endif
