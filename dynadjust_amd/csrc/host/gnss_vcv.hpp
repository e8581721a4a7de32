// GNSS variance matrices in other frames / partially scaled (host side, O(measurements) work done once in
// PrepareAdjustment).  Restates, for column-major 3k x 3k matrices with both triangles filled:
//   FormCarttoGeoRotationMatrix          include/functions/dnatemplatematrixfuncs.hpp:204-233   J = d(X,Y,Z)/d(lat,lon,h)
//   matrix_2d::sweep                     include/math/dnamatrix_contiguous.cpp:903-943          (J^-1)
//   Prpagate_Variances_Geo_Cart          dnatemplatematrixfuncs.hpp:300-313                      V <- R V R^T
//   PropagateVariances_GeoCart_Cluster   :355-365,  ScaleMatrix :368-375,  ScaleGPSVCV_Cluster :404-443
// of /root/reference/dynadjust/.  The reference forms block-diagonal 3k x 3k rotation matrices and multiplies densely;
// the zero blocks contribute nothing, so the same sums are formed here block by block.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "geodesy.hpp"

namespace dynadjust {
namespace gnssvcv {

struct Mat3 {
    double m[3][3];
};

inline Mat3 GeoToCartJacobian(double lat, double lon, double h, const geodesy::Ellipsoid& e = geodesy::Ellipsoid()) {
    const double coslat = std::cos(lat), sinlat = std::sin(lat), coslon = std::cos(lon), sinlon = std::sin(lon);
    const double term1_a = e.a * e.e2;
    const double one_minus_esq = 1.0 - e.e2;
    const double nu = e.a / std::sqrt(1.0 - e.e2 * sinlat * sinlat);
    const double nu_plus_h = nu + h;
    const double nu_1minuse2_plus_h = nu * one_minus_esq + h;
    const double term1_b = term1_a * sinlat * coslat;
    const double term1_c = std::pow(1.0 - e.e2 * sinlat * sinlat, 1.5);
    Mat3 J;
    J.m[0][0] = (term1_b * coslat * coslon / term1_c) - (nu_plus_h * sinlat * coslon);
    J.m[0][1] = -nu_plus_h * coslat * sinlon;
    J.m[0][2] = coslat * coslon;
    J.m[1][0] = (term1_b * coslat * sinlon / term1_c) - (nu_plus_h * sinlat * sinlon);
    J.m[1][1] = nu_plus_h * coslat * coslon;
    J.m[1][2] = coslat * sinlon;
    J.m[2][0] = (term1_b * one_minus_esq * sinlat / term1_c) + (nu_1minuse2_plus_h * coslat);
    J.m[2][1] = 0.0;
    J.m[2][2] = sinlat;
    return J;
}

// the sweep operator over all three pivots = inverse ("allows negative diagonal terms")
inline void Sweep(Mat3& A) {
    const double eps = 1.0e-8;
    for (int k = 0; k < 3; ++k) {
        if (std::fabs(A.m[k][k]) < eps) {
            for (int it = 0; it < 3; ++it) A.m[it][k] = A.m[k][it] = 0.0;
            continue;
        }
        const double d = 1.0 / A.m[k][k];
        A.m[k][k] = d;
        for (int i = 0; i < 3; ++i)
            if (i != k) A.m[i][k] *= -d;
        for (int j = 0; j < 3; ++j)
            if (j != k) A.m[k][j] *= d;
        for (int i = 0; i < 3; ++i)
            if (i != k)
                for (int j = 0; j < 3; ++j)
                    if (j != k) A.m[i][j] += A.m[i][k] * A.m[k][j] / d;
    }
}

// V <- R V R^T, R = blockdiag(R[0] .. R[k-1])
inline void Congruence(std::vector<double>& V, uint32_t k, const std::vector<Mat3>& R) {
    const uint32_t nc = 3 * k;
    std::vector<double> T((size_t)nc * nc);
    for (uint32_t a = 0; a < k; ++a)
        for (uint32_t col = 0; col < nc; ++col)
            for (int i = 0; i < 3; ++i) {
                double sum = 0.0;
                for (int t = 0; t < 3; ++t) sum += R[a].m[i][t] * V[(size_t)col * nc + 3 * a + t];
                T[(size_t)col * nc + 3 * a + i] = sum;
            }
    for (uint32_t b = 0; b < k; ++b)
        for (uint32_t row = 0; row < nc; ++row)
            for (int j = 0; j < 3; ++j) {
                double sum = 0.0;
                for (int t = 0; t < 3; ++t) sum += T[(size_t)(3 * b + t) * nc + row] * R[b].m[j][t];
                V[(size_t)(3 * b + j) * nc + row] = sum;
            }
}

// llh: lat, lon, h of the point each vector's rotation is formed at (3 per vector)
inline void PropagateGeoCart(std::vector<double>& V, uint32_t k, const std::vector<double>& llh, bool geo_to_cart) {
    std::vector<Mat3> R(k);
    for (uint32_t a = 0; a < k; ++a) {
        R[a] = GeoToCartJacobian(llh[3 * a], llh[3 * a + 1], llh[3 * a + 2]);
        if (!geo_to_cart) Sweep(R[a]);
    }
    Congruence(V, k, R);
}

inline void ScaleGPSVCV(std::vector<double>& V, uint32_t k, const std::vector<double>& llh, double pScale, double lScale, double hScale,
                        bool v_is_geographic) {
    const uint32_t nc = 3 * k;
    std::vector<Mat3> R(k), Ri(k);
    for (uint32_t a = 0; a < k; ++a) {
        R[a] = GeoToCartJacobian(llh[3 * a], llh[3 * a + 1], llh[3 * a + 2]);
        Ri[a] = R[a];
        Sweep(Ri[a]);
    }
    if (!v_is_geographic) Congruence(V, k, Ri);
    const double sc[3] = {std::sqrt(pScale), std::sqrt(lScale), std::sqrt(hScale)};
    for (uint32_t col = 0; col < nc; ++col)
        for (uint32_t row = 0; row < nc; ++row) V[(size_t)col * nc + row] = (sc[row % 3] * V[(size_t)col * nc + row]) * sc[col % 3];
    Congruence(V, k, R);
}

}  // namespace gnssvcv
}  // namespace dynadjust
