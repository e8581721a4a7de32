// Terrestrial measurement models of the adjustment path: computed measurement + design row (partial derivatives with
// respect to the cartesian coordinates of the measurement's stations) and the one-time reductions, for the types
//   A horizontal angle      UpdateDesignNormalMeasMatrices_A    dnaadjust.cpp:4754
//   B / K azimuths          ..._BK                              dnaadjust.cpp:4913
//   C / E / M distances     ..._C / _E / _M / _CEM              dnaadjust.cpp:5017-5080, 5242, 5400
//   S slope distance        ..._S                               dnaadjust.cpp:5437
//   V / Z zenith, vertical  ..._V / _Z                          dnaadjust.cpp:5504, 5613
//   L level difference      ..._L                               dnaadjust.cpp:5717
//   H / R heights           ..._H / _R / _HR                    dnaadjust.cpp:5969-6054
// and the geometry they call (include/functions/dnatemplategeodesyfuncs.hpp:627-1220 of /root/reference/dynadjust/).
// One source for the device kernels (adjust_kernels.hip) and for the host facade (one-time reductions in
// PrepareAdjustment, adjusted-measurement bookkeeping in GenerateStatistics): every function is __host__ __device__.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TM_HD __host__ __device__ inline
#else
#define TM_HD inline
#endif

namespace dnagpu {
namespace tm {

constexpr double PI = 3.14159265358979323846;
constexpr double TWO_PI = 2.0 * PI;
constexpr double HALF_PI = 0.5 * PI;
constexpr double GRS80_A = 6378137.0;
constexpr double GRS80_E2 = 2.0 / 298.257222101 - (1.0 / 298.257222101) * (1.0 / 298.257222101);
constexpr double E4_SEC_DEFLECTION = 0.0001 * (PI / 648000.0);   // dnaconsts.hpp:110

// station data a measurement needs besides the coordinates: the record's current geodetic position, geoid separation
// and deflections of the vertical (station_t::currentLatitude/Longitude/Height, geoidSep, verticalDef, meridianDef)
struct StationGeo {
    double lat, lon, h, geoid, defl_v, defl_m;
};

TM_HD bool single_station(char type) { return type == 'H' || type == 'R' || type == 'I' || type == 'J' || type == 'P' || type == 'Q'; }
// 'D' here is one angle of a direction set: the difference of two consecutive directions, an 'A' measurement whose
// weight is a row of the set's dense weight matrix (UpdateDesignNormalMeasMatrices_D, dnaadjust.cpp:5082)
TM_HD int station_count(char type) { return (type == 'A' || type == 'D') ? 3 : (single_station(type) ? 1 : 2); }
TM_HD bool is_terrestrial(char type) {
    switch (type) {
        case 'A': case 'B': case 'C': case 'D': case 'E': case 'H': case 'I': case 'J': case 'K': case 'L': case 'M': case 'P': case 'Q': case 'R': case 'S':
        case 'V': case 'Z': return true;
        default: return false;
    }
}

TM_HD double prime_vertical(double lat) {
    const double s = sin(lat);
    return GRS80_A / sqrt(1.0 - GRS80_E2 * (s * s));
}
TM_HD void nu_rho(double lat, double* nu, double* rho) {
    const double s = sin(lat), del = sqrt(1.0 - GRS80_E2 * (s * s));
    *nu = GRS80_A / del;
    *rho = GRS80_A * ((1.0 - GRS80_E2) / (del * del * del));
}
// atan_2 (dnatemplatecalcfuncs.hpp:350)
TM_HD double atan_2(double x, double y) {
    const double theta = atan(x / y);
    if (y < 0) return theta + PI;
    return x > 0 ? theta : theta + TWO_PI;
}
TM_HD void local_elements(const double* X1, const double* X2, double lat, double lon, double* e, double* n, double* up) {
    const double dX = X2[0] - X1[0], dY = X2[1] - X1[1], dZ = X2[2] - X1[2];
    const double sin_lat = sin(lat), cos_lat = cos(lat), sin_lon = sin(lon), cos_lon = cos(lon);
    *e = -sin_lon * dX + cos_lon * dY;
    *n = -sin_lat * cos_lon * dX - sin_lat * sin_lon * dY + cos_lat * dZ;
    if (up) *up = cos_lat * cos_lon * dX + cos_lat * sin_lon * dY + sin_lat * dZ;
}
TM_HD double direction_en(double e, double n) {
    double d = fabs(e) < fabs(n) ? atan_2(e, n) : HALF_PI - atan_2(n, e);
    if (d < 0) d += TWO_PI;
    return d;
}
TM_HD double direction(const double* X1, const double* X2, double lat, double lon, double* e, double* n) {
    local_elements(X1, X2, lat, lon, e, n, nullptr);
    return direction_en(*e, *n);
}
TM_HD void height_offset(double h, double lat, double lon, double* d) {
    d[0] = cos(lat) * cos(lon) * h;
    d[1] = cos(lat) * sin(lon) * h;
    d[2] = sin(lat) * h;
}
// local e, n, up of the line instrument -> target (ZenithDistance / VerticalAngle)
TM_HD void sight_line(const double* X1, const double* X2, const StationGeo& g1, const StationGeo& g2, double ih, double th, double* e,
                      double* n, double* up) {
    double di[3], dt[3], Xa[3] = {0.0, 0.0, 0.0}, Xb[3];
    height_offset(ih, g1.lat, g1.lon, di);
    height_offset(th, g2.lat, g2.lon, dt);
    for (int c = 0; c < 3; ++c) Xb[c] = X2[c] - X1[c] + dt[c] - di[c];
    local_elements(Xa, Xb, g1.lat, g1.lon, e, n, up);
}
TM_HD double zenith_distance(const double* X1, const double* X2, const StationGeo& g1, const StationGeo& g2, double ih, double th, double* e,
                             double* n, double* up) {
    sight_line(X1, X2, g1, g2, ih, th, e, n, up);
    return atan2(sqrt((*e) * (*e) + (*n) * (*n)), *up);
}
TM_HD double ellipsoid_height(const double* X, double lat, double* nu, double* Zn) {
    *nu = prime_vertical(lat);
    *Zn = GRS80_E2 * (*nu) * sin(lat);
    return sqrt(X[0] * X[0] + X[1] * X[1] + (X[2] + (*Zn)) * (X[2] + (*Zn))) - (*nu);
}
TM_HD double chord_distance(const double* X1, const double* X2, const StationGeo& g1, const StationGeo& g2, double* d) {
    const double nu1 = prime_vertical(g1.lat), nu2 = prime_vertical(g2.lat);
    const double s1 = nu1 / (nu1 + g1.h), s2 = nu2 / (nu2 + g2.h);
    const double Zn1 = GRS80_E2 * nu1 * sin(g1.lat), Zn2 = GRS80_E2 * nu2 * sin(g2.lat);
    const double x1 = X1[0] * s1, y1 = X1[1] * s1, z1 = (X1[2] + Zn1) * s1 - Zn1;
    const double x2 = X2[0] * s2, y2 = X2[1] * s2, z2 = (X2[2] + Zn2) * s2 - Zn2;
    d[0] = x2 - x1;
    d[1] = y2 - y1;
    d[2] = z2 - z1;
    return sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
TM_HD double radius_in_chord_direction(const double* X1, const double* X2, const StationGeo& g1, const StationGeo& g2) {
    double nu, rho, e, n;
    nu_rho((g1.lat + g2.lat) / 2.0, &nu, &rho);
    const double dir = direction(X1, X2, g1.lat, g1.lon, &e, &n);
    const double c = cos(dir), s = sin(dir);
    return rho * nu / ((nu * c * c) + (rho * s * s));
}
TM_HD double ellipsoid_arc_to_chord(double arc, const double* X1, const double* X2, const StationGeo& g1, const StationGeo& g2) {
    const double r = radius_in_chord_direction(X1, X2, g1, g2);
    return 2.0 * r * sin(arc / 2.0 / r);
}
TM_HD double ellipsoid_chord_to_arc(double chord, const double* X1, const double* X2, const StationGeo& g1, const StationGeo& g2) {
    const double r = radius_in_chord_direction(X1, X2, g1, g2);
    return asin(chord / 2.0 / r) * 2.0 * r;
}
TM_HD double msl_arc_to_ellipsoid_chord(double arc, const StationGeo& g1, const StationGeo& g2) {
    double nu, rho;
    nu_rho((g1.lat + g2.lat) / 2.0, &nu, &rho);
    const double R = sqrt(nu * rho), r = R + (g1.geoid + g2.geoid) / 2.0;
    const double msl_chord = 2.0 * r * sin(arc / 2.0 / r);
    double c = msl_chord * msl_chord;
    c -= (g2.geoid - g1.geoid) * (g2.geoid - g1.geoid);
    c /= 1.0 + g1.geoid / R;
    c /= 1.0 + g2.geoid / R;
    return sqrt(c);
}
TM_HD double ellipsoid_chord_to_msl_arc(double chord, const StationGeo& g1, const StationGeo& g2) {
    double nu, rho;
    nu_rho((g1.lat + g2.lat) / 2.0, &nu, &rho);
    const double R = sqrt(nu * rho);
    double c = chord * chord;
    c *= 1.0 + g1.geoid / R;
    c *= 1.0 + g2.geoid / R;
    c += (g2.geoid - g1.geoid) * (g2.geoid - g1.geoid);
    const double r = R + (g1.geoid + g2.geoid) / 2.0;
    return asin(sqrt(c) / 2.0 / r) * 2.0 * r;
}

// CartToGeo (dnatemplategeodesyfuncs.hpp:154-225; Lin & Wang, Newton iteration on the ellipsoid normal)
TM_HD void cart_to_geo(const double* X, double* lat, double* lon, double* h) {
    const double a_ = GRS80_A, b_ = GRS80_A * (1.0 - 1.0 / 298.257222101);
    const double x = X[0], y = X[1], z = X[2];
    const double p2 = x * x + y * y, p = sqrt(p2), a2 = a_ * a_, b2 = b_ * b_, Z2 = z * z;
    const double a2Z2 = a2 * Z2, b2p2 = b2 * p2, A = a2Z2 + b2p2;
    double m0 = (a_ * b_ * sqrt(A) * A - a2 * b2 * A) / (2. * ((a2 * a2Z2) + (b2 * b2p2)));
    double twom, a2twom, b2twom, f, df, m = m0;
    for (int i = 0; i < 5; ++i) {
        m = m0;
        twom = m * 2.;
        a2twom = a2 + twom;
        b2twom = b2 + twom;
        f = (a2 * p2 / (a2twom * a2twom)) + (b2 * Z2 / (b2twom * b2twom)) - 1.;
        if (fabs(f) < 1.0e-12) break;
        df = -4. * ((a2 * p2 / (a2twom * a2twom * a2twom)) + (b2 * Z2 / (b2twom * b2twom * b2twom)));
        m0 = m - (f / df);
        m = m0;
    }
    twom = m * 2.;
    const double p_E = a2 * p / (a2 + twom), Z_E = b2 * z / (b2 + twom);
    *lat = atan(a2 * Z_E / (b2 * p_E));
    *lon = atan(y / x);
    if (x < 0.0 && y > 0.0) *lon += PI;
    else if (x < 0.0 && y < 0.0) *lon = -(PI - *lon);
    *h = sqrt(((p - p_E) * (p - p_E)) + ((z - Z_E) * (z - Z_E)));
    if ((p + fabs(z)) < (p_E + fabs(Z_E))) *h *= -1.;
}

// The measurement as the adjustment uses it: term1, except that E and M are re-derived from the supplied arc
// (preAdjMeas) every time the design is filled (dnaadjust.cpp:5254, 5412)
TM_HD double working_value(char type, double term1, double pre_adj_meas, const double* X1, const double* X2, const StationGeo& g1,
                           const StationGeo& g2) {
    if (type == 'E') return ellipsoid_arc_to_chord(pre_adj_meas, X1, X2, g1, g2);
    if (type == 'M') return msl_arc_to_ellipsoid_chord(pre_adj_meas, g1, g2);
    return term1;
}

// computed measurement and design row (row[0..2] station 1, [3..5] station 2, [6..8] station 3)
TM_HD double evaluate(char type, const double* X1, const double* X2, const double* X3, const StationGeo& g1, const StationGeo& g2, double ih,
                      double th, double* row) {
    const double cos_lat = cos(g1.lat), sin_lat = sin(g1.lat), cos_long = cos(g1.lon), sin_long = sin(g1.lon);
    for (int i = 0; i < 9; ++i) row[i] = 0.0;
    double comp = 0.0;
    switch (type) {
        case 'A': case 'D': {
            double e12, n12, e13, n13;
            const double d12 = direction(X1, X2, g1.lat, g1.lon, &e12, &n12);
            double d13 = direction(X1, X3, g1.lat, g1.lon, &e13, &n13);
            if (d12 > d13) d13 += TWO_PI;
            comp = d13 - d12;
            const double slc = sin_lat * cos_long, sls = sin_lat * sin_long;
            const double c12 = cos(d12) * cos(d12) / (n12 * n12), c13 = cos(d13) * cos(d13) / (n13 * n13);
            row[0] = c13 * (n13 * sin_long - e13 * slc) - c12 * (n12 * sin_long - e12 * slc);
            row[1] = c13 * (-n13 * cos_long - e13 * sls) - c12 * (-n12 * cos_long - e12 * sls);
            row[2] = c13 * e13 * cos_lat - c12 * e12 * cos_lat;
            row[3] = c12 * (n12 * sin_long - e12 * slc);
            row[4] = c12 * (-n12 * cos_long - e12 * sls);
            row[5] = c12 * e12 * cos_lat;
            row[6] = -c13 * (n13 * sin_long - e13 * slc);
            row[7] = -c13 * (-n13 * cos_long - e13 * sls);
            row[8] = -c13 * e13 * cos_lat;
            break;
        }
        case 'B': case 'K': {
            double e12, n12;
            comp = direction(X1, X2, g1.lat, g1.lon, &e12, &n12);
            const double slc = sin_lat * cos_long, sls = sin_lat * sin_long;
            const double c12 = cos(comp) * cos(comp) / (n12 * n12);
            const double dx = c12 * (n12 * sin_long - e12 * slc), dy = c12 * (-n12 * cos_long - e12 * sls), dz = c12 * e12 * cos_lat;
            row[0] = dx; row[1] = dy; row[2] = dz;
            row[3] = -dx; row[4] = -dy; row[5] = -dz;
            break;
        }
        case 'C': case 'E': case 'M': {
            double d[3];
            comp = chord_distance(X1, X2, g1, g2, d);
            for (int c = 0; c < 3; ++c) {
                row[c] = -d[c] / comp;
                row[3 + c] = d[c] / comp;
            }
            break;
        }
        case 'S': {
            double di[3], dt[3], d[3];
            height_offset(ih, g1.lat, g1.lon, di);
            height_offset(th, g1.lat, g1.lon, dt);   // (sic: both offsets along station 1's normal, dnaadjust.cpp:5466)
            for (int c = 0; c < 3; ++c) d[c] = X2[c] - X1[c] + dt[c] - di[c];
            comp = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (int c = 0; c < 3; ++c) {
                row[c] = -d[c] / comp;
                row[3 + c] = d[c] / comp;
            }
            break;
        }
        case 'V': case 'Z': {
            double e, n, up, dx, dy, dz;
            const double zen = zenith_distance(X1, X2, g1, g2, ih, th, &e, &n, &up);
            const double e2n2 = e * e + n * n, se = sqrt(e2n2);
            if (type == 'V') {
                comp = zen;
                const double se2n2_up2 = se / (up * up), up_se2n2 = up * se, cos2v = cos(comp) * cos(comp);
                dx = cos2v * (((e * sin_long + n * sin_lat * cos_long) / up_se2n2) + cos_lat * cos_long * se2n2_up2);
                dy = cos2v * (((-e * cos_long + n * sin_lat * sin_long) / up_se2n2) + cos_lat * sin_long * se2n2_up2);
                dz = cos2v * ((-n * cos_lat / up_se2n2) + sin_lat * se2n2_up2);
            } else {
                comp = atan2(up, se);
                const double se_d = se / e2n2, up_d = up / (se * e2n2), cos2v = cos(comp) * cos(comp);
                dx = cos2v * ((-cos_lat * cos_long * se_d) - ((e * sin_long + n * sin_lat * cos_long) * up_d));
                dy = cos2v * ((-cos_lat * sin_long * se_d) + ((e * cos_long - n * sin_lat * sin_long) * up_d));
                dz = cos2v * ((-sin_lat * se_d) + (n * cos_lat * up_d));
            }
            row[0] = dx; row[1] = dy; row[2] = dz;
            row[3] = -dx; row[4] = -dy; row[5] = -dz;
            break;
        }
        case 'L': {
            double nu1, nu2, Zn1, Zn2;
            const double h2 = ellipsoid_height(X2, g2.lat, &nu2, &Zn2);
            const double h1 = ellipsoid_height(X1, g1.lat, &nu1, &Zn1);
            comp = h2 - h1;
            row[0] = -X1[0] / (nu1 + h1);
            row[1] = -X1[1] / (nu1 + h1);
            row[2] = -(X1[2] + Zn1) / (nu1 + h1);
            row[3] = X2[0] / (nu2 + h2);
            row[4] = X2[1] / (nu2 + h2);
            row[5] = (X2[2] + Zn2) / (nu2 + h2);
            break;
        }
        case 'H': case 'R': {
            double nu1, Zn1;
            comp = ellipsoid_height(X1, g1.lat, &nu1, &Zn1);
            row[0] = X1[0] / (nu1 + comp);
            row[1] = X1[1] / (nu1 + comp);
            row[2] = (X1[2] + Zn1) / (nu1 + comp);
            break;
        }
        case 'I': case 'P': {
            // latitude has no closed form in X, Y, Z: forward differences with a 0.1 mm step (PartialD_Latitude_F / PartialD_Latitude,
            // dnatemplategeodesyfuncs.hpp:282-320; UpdateDesignNormalMeasMatrices_IP, dnaadjust.cpp:5861)
            double lon, h;
            cart_to_geo(X1, &comp, &lon, &h);
            for (int i = 0; i < 3; ++i) {
                double Xi[3] = {X1[0], X1[1], X1[2]}, lat_i;
                Xi[i] += 1.0e-4;
                cart_to_geo(Xi, &lat_i, &lon, &h);
                row[i] = (lat_i - comp) / 1.0e-4;
            }
            break;
        }
        case 'J': case 'Q': {
            // the computed longitude is the station record's, the row  -+ xy / (x^2+y^2)^1.5 / cos|sin(longitude)
            // (UpdateDesignNormalMeasMatrices_JQ, dnaadjust.cpp:5931)
            comp = g1.lon;
            const double p2 = X1[0] * X1[0] + X1[1] * X1[1];
            const double t = X1[0] * X1[1] / pow(p2, 1.5);
            row[0] = t * -1. / cos_long;
            row[1] = t / sin_long;
            break;
        }
        default: break;
    }
    return comp;
}

// measured minus computed with the angle wrap of AddMsrtoMeasMinusComp (dnaadjust.cpp:4719)
TM_HD double meas_minus_comp(char type, double value, double comp) {
    double mmc = value - comp;
    if (type == 'A' || type == 'B' || type == 'D' || type == 'K') {
        if (mmc < -5.5) mmc += TWO_PI;
        else if (mmc > 5.5) mmc -= TWO_PI;
    }
    return mmc;
}

// One-time reduction applied when the matrices are first built: returns preAdjCorr and the reduced term1 through
// *value (deflection of the vertical: A dnaadjust.cpp:4790-4845, K :4940-4970, V :5523-5547, Z :5632-5656; geoid
// separation: L :5746-5753, H :5977-5984; I :5797, J :5828)
TM_HD double reduce(char type, double* value, const double* X1, const double* X2, const double* X3, const StationGeo& g1, const StationGeo& g2,
                    const StationGeo& g3, double ih, double th) {
    const bool defl = fabs(g1.defl_v) > E4_SEC_DEFLECTION || fabs(g1.defl_m) > E4_SEC_DEFLECTION;
    double corr = 0.0, e, n, up;
    switch (type) {
        case 'A': case 'D':
            if (defl) {
                double e12, n12, e13, n13;
                const double d12 = direction(X1, X2, g1.lat, g1.lon, &e12, &n12);
                double d13 = direction(X1, X3, g1.lat, g1.lon, &e13, &n13);
                if (d12 > d13) d13 += TWO_PI;
                const double z12 = zenith_distance(X1, X2, g1, g2, ih, th, &e, &n, &up);
                const double z13 = zenith_distance(X1, X3, g1, g3, ih, th, &e, &n, &up);
                corr = (g1.defl_m * sin(d13) - g1.defl_v * cos(d13)) / tan(z13) - (g1.defl_m * sin(d12) - g1.defl_v * cos(d12)) / tan(z12);
                *value -= corr;
            }
            break;
        case 'K':
            if (defl) {
                const double az = direction(X1, X2, g1.lat, g1.lon, &e, &n);
                const double zen = zenith_distance(X1, X2, g1, g2, ih, th, &e, &n, &up);
                corr = g1.defl_v * tan(g1.lat) + ((g1.defl_m * sin(az) - g1.defl_v * cos(az)) / tan(zen));
                *value -= corr;
            }
            break;
        case 'V': case 'Z':
            if (defl) {
                const double az = direction(X1, X2, g1.lat, g1.lon, &e, &n);
                corr = g1.defl_m * cos(az) + g1.defl_v * sin(az);
                if (type == 'V') *value += corr;
                else *value -= corr;
            }
            break;
        case 'L':
            if (fabs(g1.geoid) > 1.0e-4 || fabs(g2.geoid) > 1.0e-4) {
                corr = g2.geoid - g1.geoid;
                *value += corr;
            }
            break;
        case 'H':
            if (fabs(g1.geoid) > 1.0e-4) {
                corr = g1.geoid;
                *value += corr;
            }
            break;
        case 'I':   // astronomic -> geodetic latitude: deflection in the prime meridian (dnaadjust.cpp:5797-5804)
            if (fabs(g1.defl_m) > E4_SEC_DEFLECTION) {
                corr = g1.defl_m;
                *value -= corr;
            }
            break;
        case 'J':   // astronomic -> geodetic longitude: deflection in the prime vertical x sec(latitude) (dnaadjust.cpp:5828-5835)
            if (fabs(g1.defl_v) > E4_SEC_DEFLECTION) {
                corr = g1.defl_v / cos(g1.lat);
                *value -= corr;
            }
            break;
        default: break;
    }
    return corr;
}

}  // namespace tm
}  // namespace dnagpu
