// GRS80 geodesy used on the adjustment path (host side, O(stations)).
//   GeoToCart: include/functions/dnatemplategeodesyfuncs.hpp:78-90
//   CartToGeo: include/functions/dnatemplategeodesyfuncs.hpp:154-225 (Lin & Wang 1995, Newton iteration)
// of /root/reference/dynadjust/.
#pragma once
#include <cmath>

namespace dynadjust {
namespace geodesy {

constexpr double GRS80_A = 6378137.0;
constexpr double GRS80_INV_F = 298.257222101;

struct Ellipsoid {
    double a, inv_f, f, b, e2;
    Ellipsoid(double a_ = GRS80_A, double inv_f_ = GRS80_INV_F) : a(a_), inv_f(inv_f_) {
        f = 1.0 / inv_f;
        b = a * (1.0 - f);
        e2 = 2.0 * f - f * f;
    }
};

inline double primeVertical(const Ellipsoid& e, double lat) {
    double s = std::sin(lat);
    return e.a / std::sqrt(1.0 - e.e2 * s * s);
}

inline void GeoToCart(double lat, double lon, double h, double* X, double* Y, double* Z, const Ellipsoid& e = Ellipsoid()) {
    double nu = primeVertical(e, lat);
    *X = (nu + h) * std::cos(lat) * std::cos(lon);
    *Y = (nu + h) * std::cos(lat) * std::sin(lon);
    *Z = ((nu * (1. - e.e2)) + h) * std::sin(lat);
}

inline void CartToGeo(double x, double y, double z, double* lat, double* lon, double* h, const Ellipsoid& e = Ellipsoid()) {
    const double PI = 3.14159265358979323846;
    double p2 = x * x + y * y, p = std::sqrt(p2);
    double a2 = e.a * e.a, b2 = e.b * e.b, z2 = z * z;
    double a2z2 = a2 * z2, b2p2 = b2 * p2, A = a2z2 + b2p2;
    double m = (e.a * e.b * std::sqrt(A) * A - a2 * b2 * A) / (2. * (a2 * a2z2 + b2 * b2p2));
    for (int i = 0; i < 5; ++i) {
        double tm = 2. * m, am = a2 + tm, bm = b2 + tm;
        double fv = a2 * p2 / (am * am) + b2 * z2 / (bm * bm) - 1.;
        if (std::fabs(fv) < 1e-12) break;
        double df = -4. * (a2 * p2 / (am * am * am) + b2 * z2 / (bm * bm * bm));
        m -= fv / df;
    }
    double tm = 2. * m;
    double pE = a2 * p / (a2 + tm), zE = b2 * z / (b2 + tm);
    *lat = std::atan(a2 * zE / (b2 * pE));
    *lon = std::atan(y / x);
    if (x < 0.0 && y > 0.0)
        *lon += PI;
    else if (x < 0.0 && y < 0.0)
        *lon = -(PI - *lon);
    double dp = p - pE, dz = z - zE;
    double hh = std::sqrt(dp * dp + dz * dz);
    *h = (p + std::fabs(z) < pE + std::fabs(zE)) ? -hh : hh;
}

// rotation local (e, n, up) -> cartesian at (lat, lon): columns are the e, n, up unit vectors
inline void LocalToCartRotation(double lat, double lon, double R[3][3]) {
    double sl = std::sin(lat), cl = std::cos(lat), so = std::sin(lon), co = std::cos(lon);
    R[0][0] = -so;  R[0][1] = -sl * co;  R[0][2] = cl * co;
    R[1][0] = co;   R[1][1] = -sl * so;  R[1][2] = cl * so;
    R[2][0] = 0.0;  R[2][1] = cl;        R[2][2] = sl;
}

}  // namespace geodesy
}  // namespace dynadjust
