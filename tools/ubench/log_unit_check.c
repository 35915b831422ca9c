// Host check of log_unit() (csrc/lmc_rng.hpp): the same IEEE operations (explicit fma only) against glibc. gcc -O2 -ffp-contract=off -mfma tools/ubench/log_unit_check.c -lm
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double rcp_approx(double y) { return (double)(float)(1.0 / y); }   // stands in for v_rcp_f64 (>= 24 good bits assumed)

static double log01(double x) {   // x in (0, 1), normal
    int e;
    double m = frexp(x, &e);            // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m = m + m; e -= 1; }    // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;
    const double y = 2.0 + f;
    double r = rcp_approx(y);
    double t = fma(-y, r, 1.0); r = fma(r, t, r);
    t = fma(-y, r, 1.0); r = fma(r, t, r);
    const double s = f * r;
    const double z = s * s;
    const double w = z * z;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    // dk*ln2_hi - ((hfsq - (s*(hfsq+R) + dk*ln2_lo)) - f)
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

static double ulps(double a, double b) {
    if (a == b) return 0;
    int ex; frexp(b, &ex);
    return fabs(a - b) / ldexp(1.0, ex - 53);
}

int main() {
    uint64_t st = 88172645463325252ull;
    double worst = 0, worstf = 0, worstx = 0; long over1 = 0, n = 20000000;
    for (long i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double x = (double)(st >> 11) / 9007199254740992.0;
        if (i % 5 == 0) x = 1.0 - x * 1e-3;          // near 1
        if (i % 7 == 0) x = x * 1e-6;                // small
        if (x <= 0.0 || x >= 1.0) continue;
        const double a = log01(x), b = log(x);
        const double u = ulps(a, b);
        if (u > worst) { worst = u; worstx = x; }
        if (u > 1.0) ++over1;
        const double fa = sqrt(-2.0 * a / x), fb = sqrt(-2.0 * b / x);
        const double rel = fabs(fa - fb) / fb;
        if (rel > worstf) worstf = rel;
    }
    printf("log: worst %.3f ulp vs glibc at x=%.17g; >1 ulp: %ld of %ld; polar factor worst relative difference %.3e\n", worst, worstx, over1, n, worstf);
    return 0;
}
