// mhx_text.cpp -- the scalar, text-producing pieces around the kernels: binomial tail for
// the `mash dist` p-value column, and the `mash bounds` table (Mash 2.x CommandDistance.cpp
// pValue(), CommandBounds.cpp run()).  Their text is scraped by
// /root/reference/auriclass/classes.py:111-116 (dist TSV) and :352-375 (bounds table), so it
// has to match `ostream << double` (== printf %g) character for character.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "mhx_internal.h"

namespace mhx {

std::string fmt_g(double v)
{
    char b[64];
    snprintf(b, sizeof(b), "%g", v);
    return b;
}

// Regularised incomplete beta I_x(a, b) by the continued fraction (modified Lentz), using
// the symmetry I_x(a,b) = 1 - I_{1-x}(b,a) to stay in the rapidly converging half.
static double betacf(double a, double b, double x)
{
    const double tiny = 1e-300, eps = 1e-16;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 1000000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}

static double betainc(double a, double b, double x)
{
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double lnpre = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x);
    if (x < (a + 1.0) / (a + b + 2.0)) return exp(lnpre) * betacf(a, b, x) / a;
    return 1.0 - exp(lnpre) * betacf(b, a, 1.0 - x) / b;
}

// P[Binomial(n, p) <= x] = I_{1-p}(n - x, x + 1)
double binomial_cdf(uint64_t x, double p, uint64_t n)
{
    if (x >= n) return 1.0;
    return betainc((double)(n - x), (double)x + 1.0, 1.0 - p);
}
// P[Binomial(n, p) >= x] = I_p(x, n - x + 1)
double binomial_sf_ge(uint64_t x, double p, uint64_t n)
{
    if (x == 0) return 1.0;
    if (x > n) return 0.0;
    return betainc((double)x, (double)(n - x) + 1.0, p);
}

// `mash bounds -k K -p P`: for every (sketch size, distance) the smallest x with
// BinomialCDF(x; s, j(d)) > (1-P)/2, mapped back to a distance and reported as its
// difference to d.  The CDF is monotone in x, so the linear scan mash does is replaced by
// a bisection with the same result.
std::string bounds_text(int k, double prob)
{
    static const int sizes[] = {100, 500, 1000, 5000, 10000, 50000, 100000, 500000, 1000000};
    static const double dists[] = {0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4};
    const double q2 = (1.0 - prob) / 2.0;
    std::string o = "\nParameters (run with -h for details):\n";
    o += "   k:   " + std::to_string(k) + "\n";
    o += "   p:   " + fmt_g(prob) + "\n\n";
    for (int cont = 0; cont < 2; ++cont) {
        o += cont ? "\tScreen distance\n" : "\tMash distance\n";
        o += "Sketch";
        for (double d : dists) o += "\t" + fmt_g(d);
        o += "\n";
        for (int s : sizes) {
            o += std::to_string(s);
            for (double d : dists) {
                const double m2j = cont ? pow(1.0 - d, k) : 1.0 / (2.0 * exp(k * d) - 1.0);
                uint64_t lo = 0, hi = (uint64_t)s; // first x in [0, s] with cdf(x) > q2 (cdf(s) = 1)
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) / 2;
                    if (binomial_cdf(mid, m2j, (uint64_t)s) > q2) hi = mid; else lo = mid + 1;
                }
                const double je = (double)lo / s;
                const double j2m = cont ? 1.0 - pow(je, 1.0 / k) : -1.0 / k * log(2.0 * je / (1.0 + je));
                o += "\t" + fmt_g(j2m - d);
            }
            o += "\n";
        }
        o += "\n";
    }
    return o;
}

} // namespace mhx

using namespace mhx;

extern "C" double mhx_p_value(uint64_t common, uint64_t len_ref, uint64_t len_qry, int k, uint64_t denom)
{
    if (common == 0) return 1.0;
    const double space = pow(4.0, k);
    const double px = 1.0 / (1.0 + space / (double)len_ref);
    const double py = 1.0 / (1.0 + space / (double)len_qry);
    const double r = px * py / (px + py - px * py);
    return binomial_sf_ge(common, r, denom);
}

extern "C" int mhx_bounds(int k, double p, char *buf, size_t cap, size_t *need)
{
    clear_error();
    if (k < 1 || k > 32 || !(p > 0.0 && p < 1.0)) return fail(MHX_E_ARG, "bounds: k in 1..32 and 0 < p < 1 required");
    std::string t;
    try {
        t = bounds_text(k, p);
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_bounds: %s", e.what());
    }
    if (need) *need = t.size() + 1;
    if (cap == 0) return MHX_OK;
    if (!buf || cap < t.size() + 1) return fail(MHX_E_CAPACITY, "bounds: buffer too small (%zu needed)", t.size() + 1);
    memcpy(buf, t.c_str(), t.size() + 1);
    return MHX_OK;
}
