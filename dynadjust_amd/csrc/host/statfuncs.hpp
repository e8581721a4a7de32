// Quantiles of the normal and chi-squared distributions for the global test of the adjustment
// (the reference takes them from boost::math: dnaadjust.cpp:203-206 and :6866-6880).  Scalar host code.
#pragma once
#include <cmath>
#include <limits>

namespace dynadjust {
namespace stat {

// standard normal quantile: rational start (Acklam) polished by two Halley steps on erfc
inline double normal_quantile(double p) {
    if (!(p > 0.0 && p < 1.0)) return p <= 0.0 ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
    static const double a[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                                1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01,
                                -1.328068155288572e+01};
    static const double c[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    double x;
    if (p < 0.02425) {
        double q = std::sqrt(-2.0 * std::log(p));
        x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    } else if (p > 1.0 - 0.02425) {
        double q = std::sqrt(-2.0 * std::log(1.0 - p));
        x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    } else {
        double q = p - 0.5, r = q * q;
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
            (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
    }
    for (int it = 0; it < 2; ++it) {
        double e = 0.5 * std::erfc(-x / std::sqrt(2.0)) - p;
        double u = e * std::sqrt(2.0 * M_PI) * std::exp(0.5 * x * x);
        x -= u / (1.0 + 0.5 * x * u);
    }
    return x;
}

// regularised lower incomplete gamma P(a, x): power series below a + 1, Lentz continued fraction of Q above
inline double gamma_p(double a, double x) {
    if (x <= 0.0) return 0.0;
    const double lg = std::lgamma(a);
    if (x < a + 1.0) {
        double term = 1.0 / a, sum = term;
        for (int n = 1; n < 100000; ++n) {
            term *= x / (a + n);
            sum += term;
            if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
        }
        return sum * std::exp(-x + a * std::log(x) - lg);
    }
    const double tiny = 1e-300;
    double bb = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / bb, h = d;
    for (int n = 1; n < 100000; ++n) {
        double an = -n * (n - a);
        bb += 2.0;
        d = an * d + bb;
        if (std::fabs(d) < tiny) d = tiny;
        c = bb + an / c;
        if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}

// quantile of chi-squared(dof): Wilson-Hilferty start, safeguarded Newton on P(dof/2, q/2) = p
inline double chi_squared_quantile(double dof, double p) {
    if (!(dof > 0.0)) return std::numeric_limits<double>::quiet_NaN();
    if (p <= 0.0) return 0.0;
    if (p >= 1.0) return std::numeric_limits<double>::infinity();
    const double a = 0.5 * dof;
    double z = normal_quantile(p), t = 2.0 / (9.0 * dof);
    double q = dof * std::pow(1.0 - t + z * std::sqrt(t), 3.0);
    if (!(q > 0.0)) q = std::pow(p * a * std::exp(std::lgamma(a)), 1.0 / a) * 2.0;   // small-p limit of the series
    double lo = 0.0, hi = std::numeric_limits<double>::infinity();
    for (int it = 0; it < 200; ++it) {
        double x = 0.5 * q;
        double f = gamma_p(a, x) - p;
        if (f > 0.0) hi = q; else lo = q;
        double pdf = 0.5 * std::exp(-x + (a - 1.0) * std::log(x) - std::lgamma(a));   // d/dq P(a, q/2)
        double step = pdf > 0.0 ? f / pdf : 0.0;
        double qn = q - step;
        if (!(qn > lo && qn < hi) || pdf <= 0.0) qn = std::isinf(hi) ? 2.0 * q : 0.5 * (lo + hi);
        if (std::fabs(qn - q) <= 1e-14 * std::fabs(qn)) return qn;
        q = qn;
    }
    return q;
}

}  // namespace stat
}  // namespace dynadjust
