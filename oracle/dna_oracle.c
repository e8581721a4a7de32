/*
 * dna_oracle.c -- CPU restatement of DynAdjust's phased / simultaneous adjustment path.
 * TEST INFRASTRUCTURE ONLY (see dna_oracle.h).  Build with -ffp-contract=off.
 *
 * Reference files (under /root/reference/dynadjust/):
 *   ADJ  = dynadjust/dnaadjust/dnaadjust.cpp
 *   MAT  = include/math/dnamatrix_contiguous.cpp      MATH = ...contiguous.hpp
 *   SEG  = include/io/seg_file.cpp
 */
#define _GNU_SOURCE
#include "dna_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ORC_PROFILE=1: where an adjustment's wall time goes (printed by orc_profile_print) */
#include <time.h>
enum { T_LAPACK, T_PACK, T_SYMV, T_COPY, T_CARRY, T_RHS, T_COUNT };
static double orc_t[T_COUNT];
static const char* orc_t_name[T_COUNT] = {"dpotrf+dpotri", "pack/unpack", "N^-1 rhs (packed symv)", "matrix copies", "junction carry", "rhs formation"};
static double orc_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void orc_profile_print(void) {
    for (int i = 0; i < T_COUNT; ++i) fprintf(stderr, "oracle: %-26s %9.3f s\n", orc_t_name[i], orc_t[i]);
}
void orc_profile_reset(void) {
    for (int i = 0; i < T_COUNT; ++i) orc_t[i] = 0.0;
}

/* ========================================================================== */
/* L2: matrix_2d                                                               */
/* ========================================================================== */

/* MATH:363-365 */
size_t orc_packed_index(uint32_t n, uint32_t i, uint32_t j) {
    return (size_t)j * n - (j ? (size_t)j * (j - 1) / 2 : 0) + (i - j);
}

typedef void (*lapack_potrf_t)(const char*, const int*, double*, const int*, int*);
static lapack_potrf_t ext_potrf = NULL, ext_potri = NULL;
static void* ext_handle = NULL;
static char lapack_name[512] = "builtin";

int orc_set_lapack(const char* lib) {
    if (ext_handle) {
        dlclose(ext_handle);
        ext_handle = NULL;
    }
    ext_potrf = ext_potri = NULL;
    snprintf(lapack_name, sizeof(lapack_name), "builtin");
    if (!lib || !*lib) return 0;
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return -1;
    /* MKL / reference LAPACK export dpotrf_; the OpenBLAS that ships inside scipy prefixes its symbols */
    lapack_potrf_t f = (lapack_potrf_t)dlsym(h, "dpotrf_");
    lapack_potrf_t g = (lapack_potrf_t)dlsym(h, "dpotri_");
    if (!f || !g) {
        f = (lapack_potrf_t)dlsym(h, "scipy_dpotrf_");
        g = (lapack_potrf_t)dlsym(h, "scipy_dpotri_");
    }
    if (!f || !g) {
        dlclose(h);
        return -2;
    }
    ext_handle = h;
    ext_potrf = f;
    ext_potri = g;
    snprintf(lapack_name, sizeof(lapack_name), "%s", lib);
    return 0;
}

const char* orc_lapack_name(void) { return lapack_name; }

/* thread count of the external LAPACK (MKL_Set_Num_Threads when the MKL runtime is loaded) */
int orc_set_threads(int n) {
    if (!ext_handle || n <= 0) return -1;
    static const char* names[] = {"MKL_Set_Num_Threads", "scipy_openblas_set_num_threads", "openblas_set_num_threads", "scipy_goto_set_num_threads"};
    for (unsigned q = 0; q < sizeof(names) / sizeof(names[0]); ++q) {
        void (*setn)(int) = (void (*)(int))dlsym(ext_handle, names[q]);
        if (setn) {
            setn(n);
            return 0;
        }
    }
    return -2;
}

/* thread count for LAPACK calls made by the CALLING thread only (MKL_Set_Num_Threads_Local): the reference's --multi-thread
 * mode runs its forward and its reverse pass on two threads that each call the BLAS (dnaadjust-multi.cpp:92-244) */
int orc_set_threads_local(int n) {      /* n = 0: back to the global setting */
    if (!ext_handle || n < 0) return -1;
    int (*setl)(int) = (int (*)(int))dlsym(ext_handle, "MKL_Set_Num_Threads_Local");
    if (!setl) return -2;
    setl(n);
    return 0;
}

#define A_(i, j) a[(size_t)(j) * lda + (i)]

/* dpotrf('L'), unblocked left-looking column Cholesky (LAPACK dpotf2 lower) */
static int builtin_potrf_lower(uint32_t n, double* a, uint32_t lda) {
    for (uint32_t j = 0; j < n; ++j) {
        double ajj = A_(j, j);
        for (uint32_t k = 0; k < j; ++k) ajj -= A_(j, k) * A_(j, k);
        if (!(ajj > 0.0)) {
            A_(j, j) = ajj;
            return (int)j + 1;
        }
        ajj = sqrt(ajj);
        A_(j, j) = ajj;
        /* a[j+1:, j] -= A[j+1:, 0:j] * a[j, 0:j]^T  (column sweeps: contiguous) */
        for (uint32_t k = 0; k < j; ++k) {
            double ajk = A_(j, k);
            if (ajk == 0.0) continue;
            double* col = &A_(0, k);
            double* dst = &A_(0, j);
            for (uint32_t i = j + 1; i < n; ++i) dst[i] -= col[i] * ajk;
        }
        double r = 1.0 / ajj;
        for (uint32_t i = j + 1; i < n; ++i) A_(i, j) *= r;
    }
    return 0;
}

/* dpotri('L') = dtrti2('L','N') followed by dlauu2('L') */
static int builtin_potri_lower(uint32_t n, double* a, uint32_t lda) {
    /* dtrti2 lower, non-unit: columns from last to first */
    for (uint32_t jj = n; jj-- > 0;) {
        if (A_(jj, jj) == 0.0) return (int)jj + 1;
        A_(jj, jj) = 1.0 / A_(jj, jj);
        double ajj = -A_(jj, jj);
        if (jj + 1 < n) {
            /* x = L22^-1 (already inverted, lower) * a[jj+1:, jj]   (dtrmv lower, no-trans) */
            for (uint32_t i = n; i-- > jj + 1;) {
                double s = 0.0;
                for (uint32_t k = jj + 1; k <= i; ++k) s += A_(i, k) * A_(k, jj);
                A_(i, jj) = s;  /* rows below i are already final; row i only needs k <= i of the old vector */
            }
            for (uint32_t i = jj + 1; i < n; ++i) A_(i, jj) *= ajj;
        }
    }
    /* dlauu2 lower: A = L^T L */
    for (uint32_t i = 0; i < n; ++i) {
        double aii = A_(i, i);
        if (i + 1 < n) {
            double d = 0.0;
            for (uint32_t k = i; k < n; ++k) d += A_(k, i) * A_(k, i);
            /* row i, columns 0..i-1:  a(i,j) = aii*a(i,j) + sum_{k>i} a(k,i) a(k,j) */
            for (uint32_t j = 0; j < i; ++j) {
                double s = aii * A_(i, j);
                for (uint32_t k = i + 1; k < n; ++k) s += A_(k, i) * A_(k, j);
                A_(i, j) = s;
            }
            A_(i, i) = d;
        } else {
            for (uint32_t j = 0; j <= i; ++j) A_(i, j) *= aii;
        }
    }
    return 0;
}

int orc_potrf_lower(uint32_t n, double* a, uint32_t lda) {
    if (ext_potrf) {
        int nn = (int)n, ld = (int)lda, info = 0;
        ext_potrf("L", &nn, a, &ld, &info);
        return info;
    }
    return builtin_potrf_lower(n, a, lda);
}

int orc_potri_lower(uint32_t n, double* a, uint32_t lda) {
    if (ext_potri) {
        int nn = (int)n, ld = (int)lda, info = 0;
        ext_potri("L", &nn, a, &ld, &info);
        return info;
    }
    return builtin_potri_lower(n, a, lda);
}
#undef A_

/* MAT:993-1019 (non-packed path, LOWER_IS_CLEARED = false, fillupper) */
int orc_cholesky_inverse_full(double* a, uint32_t n, uint32_t lda) {
    if (n < 1) return 0;
    int info = orc_potrf_lower(n, a, lda);
    if (info) return info;
    info = orc_potri_lower(n, a, lda);
    if (info) return info;
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t i = j + 1; i < n; ++i) a[(size_t)i * lda + j] = a[(size_t)j * lda + i];
    return 0;
}

/* MAT:962-991 */
int orc_cholesky_inverse_packed(double* ap, uint32_t n) {
    if (n < 1) return 0;
    double* full = (double*)malloc((size_t)n * n * sizeof(double));
    if (!full) return -1;
    /* column j of the packed lower triangle (rows j..n-1) is contiguous in both layouts */
    double t0 = orc_now();
    for (uint32_t j = 0; j < n; ++j) memcpy(full + (size_t)j * n + j, ap + orc_packed_index(n, j, j), (size_t)(n - j) * sizeof(double));
    double t1 = orc_now();
    int info = orc_potrf_lower(n, full, n);
    if (!info) info = orc_potri_lower(n, full, n);
    double t2 = orc_now();
    if (!info)
        for (uint32_t j = 0; j < n; ++j) memcpy(ap + orc_packed_index(n, j, j), full + (size_t)j * n + j, (size_t)(n - j) * sizeof(double));
    free(full);
    orc_t[T_PACK] += (t1 - t0) + (orc_now() - t2);
    orc_t[T_LAPACK] += t2 - t1;
    return info;
}

/* MAT:1145-1152 */
void orc_scale_symmetric_diagonal_packed(double* ap, uint32_t n, const double* diag) {
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t i = j; i < n; ++i) ap[orc_packed_index(n, i, j)] *= diag[i] * diag[j];
}

/* MAT:1471-1497: cblas_dspmv(ColMajor, Lower) restated: y = A x */
void orc_multiply_sym_packed(const double* ap, const double* x, double* y, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) y[i] = 0.0;
    size_t kk = 0;
    for (uint32_t j = 0; j < n; ++j) {
        double t1 = x[j], t2 = 0.0;
        y[j] += t1 * ap[kk];
        size_t k = kk + 1;
        for (uint32_t i = j + 1; i < n; ++i, ++k) {
            y[i] += t1 * ap[k];
            t2 += ap[k] * x[i];
        }
        y[j] += t2;
        kk += n - j;
    }
}

/* ADJ:6614-6645 around ADJ:8472 (FormInverseVarianceMatrix: 1x1 special case ADJ:8474) */
int orc_inverse_normals_packed(double* ap, uint32_t n, int scale_to_unity) {
    if (n == 1) {
        ap[0] = 1.0 / ap[0];
        return 0;
    }
    double* s = NULL;
    if (scale_to_unity) {
        s = (double*)malloc((size_t)n * sizeof(double));
        for (uint32_t i = 0; i < n; ++i) s[i] = 1.0 / sqrt(ap[orc_packed_index(n, i, i)]);
        orc_scale_symmetric_diagonal_packed(ap, n, s);
    }
    int info = orc_cholesky_inverse_packed(ap, n);
    if (!info && scale_to_unity) orc_scale_symmetric_diagonal_packed(ap, n, s);
    free(s);
    return info;
}

/* ========================================================================== */
/* geodesy: GeoToCart (dnatemplategeodesyfuncs.hpp:78-90), GRS80 ellipsoid      */
/* ========================================================================== */
void orc_geo_to_cart(double lat, double lon, double h, double* x, double* y, double* z) {
    const double a = 6378137.0, inv_f = 298.257222101;
    const double f = 1.0 / inv_f;
    const double e2 = 2.0 * f - f * f;
    double s = sin(lat);
    double nu = a / sqrt(1.0 - e2 * s * s);
    *x = (nu + h) * cos(lat) * cos(lon);
    *y = (nu + h) * cos(lat) * sin(lon);
    *z = ((nu * (1.0 - e2)) + h) * sin(lat);
}

/* ========================================================================== */
/* GNSS variance matrices in other frames / partially scaled                    */
/*   FormCarttoGeoRotationMatrix (dnatemplatematrixfuncs.hpp:204-233): Jacobian */
/*   d(X,Y,Z)/d(lat,lon,h); matrix_2d::sweep (dnamatrix_contiguous.cpp:903-943);*/
/*   Prpagate_Variances_Geo_Cart (:300-313); ScaleMatrix (:368-375);            */
/*   ScaleGPSVCV_Cluster (:404-443); PropagateVariances_GeoCart_Cluster (:355). */
/* The reference builds block-diagonal 3k x 3k rotation matrices and multiplies  */
/* densely; the products with the zero blocks vanish, so the same sums are done  */
/* here block by block.  V is column-major 3k x 3k, both triangles.             */
/* ========================================================================== */
static void geo_cart_jacobian(double lat, double lon, double h, double J[3][3]) {
    const double a = 6378137.0, inv_f = 298.257222101;
    const double f = 1.0 / inv_f;
    const double e2 = 2.0 * f - f * f;
    double coslat = cos(lat), sinlat = sin(lat), coslon = cos(lon), sinlon = sin(lon);
    double term1_a = a * e2;
    double one_minus_esq = 1.0 - e2;
    double nu = a / sqrt(1.0 - e2 * sinlat * sinlat);
    double nu_plus_h = nu + h;
    double nu_1minuse2_plus_h = nu * one_minus_esq + h;
    double term1_b = term1_a * sinlat * coslat;
    double term1_c = pow(1.0 - e2 * sinlat * sinlat, 1.5);
    J[0][0] = (term1_b * coslat * coslon / term1_c) - (nu_plus_h * sinlat * coslon);
    J[0][1] = -nu_plus_h * coslat * sinlon;
    J[0][2] = coslat * coslon;
    J[1][0] = (term1_b * coslat * sinlon / term1_c) - (nu_plus_h * sinlat * sinlon);
    J[1][1] = nu_plus_h * coslat * coslon;
    J[1][2] = coslat * sinlon;
    J[2][0] = (term1_b * one_minus_esq * sinlat / term1_c) + (nu_1minuse2_plus_h * coslat);
    J[2][1] = 0.0;
    J[2][2] = sinlat;
}

/* matrix_2d::sweep(0, 3) on one 3x3 block (the off-block elements of the reference's big matrix stay zero) */
static void sweep3(double A[3][3]) {
    const double eps = 1.0e-8;
    for (int k = 0; k < 3; ++k) {
        if (fabs(A[k][k]) < eps) {
            for (int it = 0; it < 3; ++it) A[it][k] = A[k][it] = 0.0;
        } else {
            double d = 1.0 / A[k][k];
            A[k][k] = d;
            for (int i = 0; i < 3; ++i)
                if (i != k) A[i][k] *= -d;
            for (int j = 0; j < 3; ++j)
                if (j != k) A[k][j] *= d;
            for (int i = 0; i < 3; ++i)
                if (i != k)
                    for (int j = 0; j < 3; ++j)
                        if (j != k) A[i][j] += A[i][k] * A[k][j] / d;
        }
    }
}

/* V <- R V R^T with R = blockdiag(R_0 .. R_{k-1}) */
static void congruence_blocks(double* V, uint32_t k, double (*R)[3][3]) {
    const uint32_t nc = 3 * k;
    double* T = (double*)malloc((size_t)nc * nc * sizeof(double));
    for (uint32_t a = 0; a < k; ++a)          /* T = R V */
        for (uint32_t col = 0; col < nc; ++col)
            for (int i = 0; i < 3; ++i) {
                double sum = 0.0;
                for (int t = 0; t < 3; ++t) sum += R[a][i][t] * V[(size_t)col * nc + 3 * a + t];
                T[(size_t)col * nc + 3 * a + i] = sum;
            }
    for (uint32_t b = 0; b < k; ++b)          /* V = T R^T */
        for (uint32_t row = 0; row < nc; ++row)
            for (int j = 0; j < 3; ++j) {
                double sum = 0.0;
                for (int t = 0; t < 3; ++t) sum += T[(size_t)(3 * b + t) * nc + row] * R[b][j][t];
                V[(size_t)(3 * b + j) * nc + row] = sum;
            }
    free(T);
}

/* PropagateVariances_GeoCart_Cluster: V (geographic: rad, rad, m) -> cartesian (geo_to_cart != 0) or back;
 * llh: lat, lon, h of the position the rotation is formed at, 3 per vector */
void orc_propagate_geo_cart(double* V, uint32_t k, const double* llh, int geo_to_cart) {
    double(*R)[3][3] = (double(*)[3][3])malloc((size_t)k * sizeof(double[3][3]));
    for (uint32_t a = 0; a < k; ++a) {
        geo_cart_jacobian(llh[3 * a], llh[3 * a + 1], llh[3 * a + 2], R[a]);
        if (!geo_to_cart) sweep3(R[a]);
    }
    congruence_blocks(V, k, R);
    free(R);
}

/* ScaleGPSVCV_Cluster: phi / lambda / height variance scalars applied in the geographic frame.
 * v_is_geographic != 0: V is already geographic (a Y cluster given in LLH), no initial propagation. */
void orc_scale_gps_vcv(double* V, uint32_t k, const double* llh, double pScale, double lScale, double hScale, int v_is_geographic) {
    const uint32_t nc = 3 * k;
    double(*R)[3][3] = (double(*)[3][3])malloc((size_t)k * sizeof(double[3][3]));
    double(*Ri)[3][3] = (double(*)[3][3])malloc((size_t)k * sizeof(double[3][3]));
    for (uint32_t a = 0; a < k; ++a) {
        geo_cart_jacobian(llh[3 * a], llh[3 * a + 1], llh[3 * a + 2], R[a]);
        memcpy(Ri[a], R[a], sizeof(double[3][3]));
        sweep3(Ri[a]);
    }
    if (!v_is_geographic) congruence_blocks(V, k, Ri);
    /* ScaleMatrix: V <- S V S^T, S = diag(sqrt(p), sqrt(l), sqrt(h), ...) */
    const double sc[3] = {sqrt(pScale), sqrt(lScale), sqrt(hScale)};
    for (uint32_t col = 0; col < nc; ++col)
        for (uint32_t row = 0; row < nc; ++row) V[(size_t)col * nc + row] = (sc[row % 3] * V[(size_t)col * nc + row]) * sc[col % 3];
    congruence_blocks(V, k, R);
    free(R);
    free(Ri);
}

/* ========================================================================== */
/* measurement weights: ADJ:4214-4309 -> ADJ:8472 (dpotrf 'U' + dpotri 'U', 3x3) */
/* The formulas below are the 3x3 instance of U^T U factorisation, inversion of */
/* U and U^-1 U^-T, written in the same operation order as the device kernel.    */
/* ========================================================================== */
int orc_weight_3x3(const double* v, double* w) {
    double v11 = v[0], v12 = v[1], v22 = v[2], v13 = v[3], v23 = v[4], v33 = v[5];
    double u11 = sqrt(v11);
    double u12 = v12 / u11;
    double u13 = v13 / u11;
    double d22 = v22 - u12 * u12;
    double u22 = sqrt(d22);
    double u23 = (v23 - u12 * u13) / u22;
    double d33 = (v33 - u13 * u13) - u23 * u23;
    double u33 = sqrt(d33);
    if (!(v11 > 0.0) || !(d22 > 0.0) || !(d33 > 0.0)) return 1;
    double t11 = 1.0 / u11, t22 = 1.0 / u22, t33 = 1.0 / u33;
    double t12 = -(t11 * u12) * t22;
    double t23 = -(t22 * u23) * t33;
    double t13 = -(t11 * (u12 * t23 + u13 * t33));
    w[0] = (t11 * t11 + t12 * t12) + t13 * t13;
    w[1] = t12 * t22 + t13 * t23;
    w[2] = t22 * t22 + t23 * t23;
    w[3] = t13 * t33;
    w[4] = t23 * t33;
    w[5] = t33 * t33;
    return 0;
}

static inline int sym6(int i, int j) {
    int lo = i < j ? i : j, hi = i < j ? j : i;
    return hi * (hi + 1) / 2 + lo;
}

/* ========================================================================== */
/* adjustment                                                                  */
/* ========================================================================== */
typedef struct {
    uint32_t k;          /* stations in the set */
    uint32_t* stn;       /* global ids (junction order) */
    double* W;           /* packed 3k x 3k: copy of the inverted junction variances (what the
                            reference copies into the grown AtVinv columns) */
    uint32_t row0;       /* first row in b */
} pseudo_set;

typedef struct {
    uint32_t n_stn, n;
    uint32_t* stations;          /* v_parameterStationList_ (ascending), ADJ:10477-10480 */
    uint8_t *first_fwd, *first_rev;
    int first, last, isolated;   /* blockMeta_t, ADJ:10449-10474 */
    uint32_t n_jsl;
    const uint32_t* jsl;         /* v_JSL_[block], file order */
    uint32_t n_cml;
    const uint32_t* cml;
    uint32_t m;                  /* design rows of real measurements */
    double *N, *NR;              /* v_normals_, v_normalsR_ (packed) */
    double *est, *orig, *rig;    /* v_estimatedStations_, v_originalStations_, v_rigorousStations_ */
    double* rigvar;              /* v_rigorousVariances_ (packed) */
    double *corr, *corrR;        /* v_corrections_, v_correctionsR_ */
    double* b;                   /* v_measMinusComp_ (grown capacity) */
    uint32_t b_rows;
    pseudo_set ps[2];
    uint32_t n_ps;
    double *jvar, *jvarFwd;      /* v_junctionVariances_, v_junctionVariancesFwd_ (packed 3*n_jsl) */
    double *jestFwd;             /* v_junctionEstimatesFwd_[block]     (3*n_jsl) */
    double *jestRev;             /* v_junctionEstimatesRev_[block]     (3*|JSL(block-1)|) */
    double* prec;                /* v_precAdjMsrsFull_ (6 per vector, 1 per terrestrial measurement) */
    double* trow;                /* design rows of the block's terrestrial measurements (9 each, CML order), from compute_b */
    uint32_t n_trow;
} blk_t;

struct orc_adjustment {
    orc_network net;
    orc_settings set;
    int phased;
    uint32_t n_blocks;
    blk_t* blk;
    double* W;                   /* 6 per vector: its own weight block (exact path for single baselines) */
    uint32_t n_clusters;
    uint32_t* cl_off;            /* vectors of cluster c: cl_off[c] .. cl_off[c+1]-1 */
    double** cl_W;               /* dense 3k x 3k inverse variance matrix per cluster (NULL for single baselines) */
    uint32_t* simul_stations;    /* 0..n-1 for simultaneous mode */
    uint32_t* simul_cml;
    double var_C, var_F;
    uint32_t iterations;
    double max_corr_hist[64];
    double maxCorr;
    uint64_t solves;
    double sum_n3;
    double* msr_field[7];        /* per vector component, see orc_adjust_msr_field */
    /* terrestrial measurements */
    uint32_t n_tm;
    double* t_val;               /* term1: the measurement after its one-time reductions (E, M: re-derived every evaluation) */
    double* t_pre;               /* preAdjMeas: as supplied */
    double* t_corr;              /* preAdjCorr */
    double* geo;                 /* lat, lon, h per station: the bst "current" geodetic coordinates */
    double* tm_field[8];
    /* UpdateIterationDiagnostics (ADJ:7450): corrPrev_ / stnOscCount_ / oscHistory_ (dnaadjust.hpp:1274-1288), per station of the network */
    double* osc_prev;
    uint8_t* osc_seen;
    uint32_t* osc_cnt;
    int32_t* osc_slot;           /* station -> record in osc_hist, -1 = none */
    orc_osc_record* osc_hist;
    uint32_t n_osc;
    char err[512];
};

static size_t psize(uint32_t n) { return (size_t)n * (n + 1) / 2; }

/* v_blockStationsMap_.at(block)[stn] (std::map in the reference, ADJ:10498) */
static uint32_t local_index(const blk_t* B, uint32_t stn) {
    uint32_t lo = 0, hi = B->n_stn;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (B->stations[mid] < stn)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo; /* caller guarantees membership (the reference does not check either, ADJ:1072) */
}

/* matrix_2d::lower_add on a packed matrix, MATH:405-411 */
static inline void lower_add(double* N, uint32_t n, uint32_t r, uint32_t c, double v) {
    if (r < c) return;
    N[orc_packed_index(n, r, c)] += v;
}

static inline double packed_get(const double* P, uint32_t n, uint32_t i, uint32_t j) {
    if (i < j) {
        uint32_t t = i;
        i = j;
        j = t;
    }
    return P[orc_packed_index(n, i, j)];
}

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : x > y;
}

/* UpdateNormals_G (ADJ:1664-1684) via add_normal_3x3_from_atvinv_columns (ADJ:1478-1491):
 * AtVinv(stn1 rows) = -W, AtVinv(stn2 rows) = +W (ADJ:5388-5389). */
static void update_normals_G(double* N, uint32_t n, uint32_t s1, uint32_t s2, const double* w6) {
    /* station 2, scale 1:  N[s2+r][s2+c] += 1 * (+W[r][c]) */
    for (int col = 0; col < 3; ++col)
        for (int r = 0; r < 3; ++r) lower_add(N, n, s2 + r, s2 + col, 1. * w6[sym6(r, col)]);
    /* station 1, scale -1: N[s1+r][s1+c] += -1 * (-W[r][c]) */
    for (int col = 0; col < 3; ++col)
        for (int r = 0; r < 3; ++r) lower_add(N, n, s1 + r, s1 + col, -1. * (-w6[sym6(r, col)]));
    /* (stn1, stn2), scale 1:  += 1 * (-W) */
    for (int col = 0; col < 3; ++col)
        for (int r = 0; r < 3; ++r) lower_add(N, n, s1 + r, s2 + col, 1. * (-w6[sym6(r, col)]));
    /* (stn2, stn1), scale -1: += -1 * (+W) */
    for (int col = 0; col < 3; ++col)
        for (int r = 0; r < 3; ++r) lower_add(N, n, s2 + r, s1 + col, -1. * w6[sym6(r, col)]);
}

/* A cluster of k vectors ('X' baseline cluster ADJ:6056-6246 / 1687-1787, 'Y' point cluster ADJ:6249-6566 / 1790-1832):
 * AtVinv rows of every station of the cluster are accumulated first (blockadd of the weight rows with the sign of the
 * design element), then N(s, s') += AtVinv(s, cluster columns) * A(cluster rows, s').  rhs != NULL: rhs(s) += AtVinv(s,:) b. */
static void cluster_contribution(orc_adjustment* a, blk_t* B, uint32_t c, double* N, double* rhs, const double* bvec) {
    const uint32_t v0 = a->cl_off[c], k = a->cl_off[c + 1] - v0, nc = 3 * k;
    const double* W = a->cl_W[c];
    /* unique stations of the cluster (block-local indices) */
    uint32_t* st = (uint32_t*)malloc(2 * k * sizeof(uint32_t));
    uint32_t ns = 0;
    for (uint32_t j = 0; j < k; ++j) {
        uint32_t ends[2] = {a->net.stn1[v0 + j], a->net.stn2[v0 + j]};
        for (int e = 0; e < 2; ++e) {
            if (ends[e] == 0xffffffffu) continue;
            uint32_t l = local_index(B, ends[e]);
            uint32_t q = 0;
            while (q < ns && st[q] != l) ++q;
            if (q == ns) st[ns++] = l;
        }
    }
    double* AtV = (double*)calloc((size_t)ns * 3 * nc, sizeof(double)); /* row-major [station row][cluster column] */
    int* sgn = (int*)calloc((size_t)ns * k, sizeof(int));               /* design element of vector j at station q */
    for (uint32_t j = 0; j < k; ++j) {
        uint32_t ends[2] = {a->net.stn1[v0 + j], a->net.stn2[v0 + j]};
        for (int e = 0; e < 2; ++e) {
            if (ends[e] == 0xffffffffu) continue;
            uint32_t l = local_index(B, ends[e]), q = 0;
            while (st[q] != l) ++q;
            double sg = e == 0 ? -1.0 : 1.0;
            sgn[q * k + j] += (int)sg;
            for (int r = 0; r < 3; ++r)
                for (uint32_t col = 0; col < nc; ++col) AtV[((size_t)q * 3 + r) * nc + col] += sg * W[(size_t)col * nc + 3 * j + r];
        }
    }
    for (uint32_t q = 0; q < ns; ++q) {
        if (N)
            for (uint32_t p = 0; p < ns; ++p)
                for (int r = 0; r < 3; ++r)
                    for (int cc = 0; cc < 3; ++cc) {
                        double sum = 0.0;
                        for (uint32_t j = 0; j < k; ++j)
                            if (sgn[p * k + j]) sum += AtV[((size_t)q * 3 + r) * nc + 3 * j + cc] * (double)sgn[p * k + j];
                        lower_add(N, B->n, 3 * st[q] + r, 3 * st[p] + cc, sum);
                    }
        if (rhs)
            for (int r = 0; r < 3; ++r) {
                double sum = 0.0;
                for (uint32_t col = 0; col < nc; ++col) sum += AtV[((size_t)q * 3 + r) * nc + col] * bvec[col];
                rhs[3 * st[q] + r] += sum;
            }
    }
    free(AtV);
    free(sgn);
    free(st);
}

/* ========================================================================== */
/* terrestrial measurements: UpdateDesignNormalMeasMatrices_A/_BK/_CEM/_E/_M/_S/ */
/* _V/_Z/_L/_H/_HR/_R (ADJ:4754-6054) and the geometry they call                */
/* (include/functions/dnatemplategeodesyfuncs.hpp:627-1220)                      */
/* ========================================================================== */
#define ORC_PI 3.14159265358979323846
#define ORC_TWO_PI (2.0 * ORC_PI)
#define ORC_HALF_PI (0.5 * ORC_PI)
#define ORC_E4_SEC_DEFLECTION (0.0001 * (ORC_PI / 648000.0)) /* dnaconsts.hpp:110 */

static void grs80(double* a_, double* e2_) {
    const double inv_f = 298.257222101, f = 1.0 / inv_f;
    *a_ = 6378137.0;
    *e2_ = 2.0 * f - f * f;
}
static double prime_vertical(double lat) {
    double a_, e2;
    grs80(&a_, &e2);
    return a_ / sqrt(1.0 - e2 * (sin(lat) * sin(lat)));
}
static void nu_rho(double lat, double* nu, double* rho) {
    double a_, e2;
    grs80(&a_, &e2);
    double del = sqrt(1.0 - e2 * (sin(lat) * sin(lat)));
    *nu = a_ / del;
    *rho = a_ * ((1.0 - e2) / (del * del * del));
}
/* atan_2 (dnatemplatecalcfuncs.hpp:350) */
static double atan_2(double x, double y) {
    double theta = atan(x / y);
    if (y < 0) return theta + ORC_PI;
    return x > 0 ? theta : theta + ORC_TWO_PI;
}
static void local_elements(const double* X1, const double* X2, double lat, double lon, double* e, double* n, double* up) {
    double dX = X2[0] - X1[0], dY = X2[1] - X1[1], dZ = X2[2] - X1[2];
    double sin_lat = sin(lat), cos_lat = cos(lat), sin_lon = sin(lon), cos_lon = cos(lon);
    *e = -sin_lon * dX + cos_lon * dY;
    *n = -sin_lat * cos_lon * dX - sin_lat * sin_lon * dY + cos_lat * dZ;
    if (up) *up = cos_lat * cos_lon * dX + cos_lat * sin_lon * dY + sin_lat * dZ;
}
/* Direction (geodesyfuncs:679-720) */
static double direction_en(double e, double n) {
    double d = fabs(e) < fabs(n) ? atan_2(e, n) : ORC_HALF_PI - atan_2(n, e);
    if (d < 0) d += ORC_TWO_PI;
    return d;
}
static double direction(const double* X1, const double* X2, double lat, double lon, double* e, double* n) {
    local_elements(X1, X2, lat, lon, e, n, NULL);
    return direction_en(*e, *n);
}
static void height_offset(double h, double lat, double lon, double* d) { /* CartesianElementsFromInstrumentHeight */
    d[0] = cos(lat) * cos(lon) * h;
    d[1] = cos(lat) * sin(lon) * h;
    d[2] = sin(lat) * h;
}
/* local e, n, up of the line instrument -> target (ZenithDistance / VerticalAngle, geodesyfuncs:786-907) */
static void sight_line(const double* X1, const double* X2, double lat1, double lon1, double lat2, double lon2, double ih, double th,
                       double* e, double* n, double* up) {
    double di[3], dt[3], Xa[3] = {0, 0, 0}, Xb[3];
    height_offset(ih, lat1, lon1, di);
    height_offset(th, lat2, lon2, dt);
    for (int c = 0; c < 3; ++c) Xb[c] = X2[c] - X1[c] + dt[c] - di[c];
    local_elements(Xa, Xb, lat1, lon1, e, n, up);
}
static double zenith_distance(const double* X1, const double* X2, double lat1, double lon1, double lat2, double lon2, double ih, double th,
                              double* e, double* n, double* up) {
    sight_line(X1, X2, lat1, lon1, lat2, lon2, ih, th, e, n, up);
    return atan2(sqrt((*e) * (*e) + (*n) * (*n)), *up);
}
static double ellipsoid_height(const double* X, double lat, double* nu, double* Zn) { /* geodesyfuncs:909 */
    double a_, e2;
    grs80(&a_, &e2);
    *nu = prime_vertical(lat);
    *Zn = e2 * (*nu) * sin(lat);
    return sqrt(X[0] * X[0] + X[1] * X[1] + (X[2] + (*Zn)) * (X[2] + (*Zn))) - (*nu);
}
static double chord_distance(const double* X1, const double* X2, double lat1, double lat2, double h1, double h2, double* d) { /* :957 */
    double a_, e2;
    grs80(&a_, &e2);
    double nu1 = prime_vertical(lat1), nu2 = prime_vertical(lat2);
    double s1 = nu1 / (nu1 + h1), s2 = nu2 / (nu2 + h2);
    double Zn1 = e2 * nu1 * sin(lat1), Zn2 = e2 * nu2 * sin(lat2);
    double x1 = X1[0] * s1, y1 = X1[1] * s1, z1 = (X1[2] + Zn1) * s1 - Zn1;
    double x2 = X2[0] * s2, y2 = X2[1] * s2, z2 = (X2[2] + Zn2) * s2 - Zn2;
    d[0] = x2 - x1;
    d[1] = y2 - y1;
    d[2] = z2 - z1;
    return sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
static double radius_in_chord_direction(const double* X1, const double* X2, double lat1, double lon1, double lat2) { /* :983 */
    double nu, rho, e, n;
    nu_rho((lat1 + lat2) / 2.0, &nu, &rho);
    double dir = direction(X1, X2, lat1, lon1, &e, &n);
    double c = cos(dir), sn = sin(dir);
    return rho * nu / ((nu * c * c) + (rho * sn * sn));
}
static double msl_arc_to_ellipsoid_chord(double arc, double lat1, double lat2, double N1, double N2) { /* :1100, :1061, :1086 */
    double nu, rho;
    nu_rho((lat1 + lat2) / 2.0, &nu, &rho);
    double r = sqrt(nu * rho) + (N1 + N2) / 2.0;
    double msl_chord = 2.0 * r * sin(arc / 2.0 / r);
    double c = msl_chord * msl_chord;
    c -= (N2 - N1) * (N2 - N1);
    double R = sqrt(nu * rho);
    c /= 1.0 + N1 / R;
    c /= 1.0 + N2 / R;
    return sqrt(c);
}
static double ellipsoid_chord_to_msl_arc(double chord, double lat1, double lat2, double N1, double N2) { /* :1135 */
    double nu, rho;
    nu_rho((lat1 + lat2) / 2.0, &nu, &rho);
    double R = sqrt(nu * rho);
    double c = chord * chord;
    c *= 1.0 + N1 / R;
    c *= 1.0 + N2 / R;
    c += (N2 - N1) * (N2 - N1);
    double msl_chord = sqrt(c);
    double r = R + (N1 + N2) / 2.0;
    return asin(msl_chord / 2.0 / r) * 2.0 * r;
}

static int tm_station_count(char type) {
    switch (type) {
        case 'A': case 'D': return 3;   /* 'D': one angle of a direction set, see dset_* below */
        case 'H': case 'R': case 'I': case 'J': case 'P': case 'Q': return 1;
        default: return 2;
    }
}

static void cart_to_geo(const double* X, double* lat, double* lon, double* h);

/* computed measurement and design row at the cartesian coordinates X1..X3 with the current geodetic station data.
 * For E and M the ellipsoid chord equivalent of the supplied arc is (re)derived into a->t_val first (ADJ:5254, ADJ:5412). */
static void tm_evaluate(orc_adjustment* a, uint32_t t, const double* X1, const double* X2, const double* X3, double* comp, double* row) {
    const orc_network* net = &a->net;
    const char type = net->t_type[t];
    const uint32_t g1 = net->t_stn[3 * (size_t)t], g2 = net->t_stn[3 * (size_t)t + 1];
    const double lat1 = a->geo[3 * (size_t)g1], lon1 = a->geo[3 * (size_t)g1 + 1];
    const double cos_lat = cos(lat1), sin_lat = sin(lat1), cos_long = cos(lon1), sin_long = sin(lon1);
    for (int i = 0; i < 9; ++i) row[i] = 0.0;
    switch (type) {
        case 'A': case 'D': {
            double e12, n12, e13, n13;
            double d12 = direction(X1, X2, lat1, lon1, &e12, &n12);
            double d13 = direction(X1, X3, lat1, lon1, &e13, &n13);
            if (d12 > d13) d13 += ORC_TWO_PI;                                     /* HorizontalAngle, :733 */
            *comp = d13 - d12;
            double slc = sin_lat * cos_long, sls = sin_lat * sin_long;
            double c12 = cos(d12) * cos(d12) / (n12 * n12), c13 = cos(d13) * cos(d13) / (n13 * n13);
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
            *comp = direction(X1, X2, lat1, lon1, &e12, &n12);
            double slc = sin_lat * cos_long, sls = sin_lat * sin_long;
            double c12 = cos(*comp) * cos(*comp) / (n12 * n12);
            double dx = c12 * (n12 * sin_long - e12 * slc), dy = c12 * (-n12 * cos_long - e12 * sls), dz = c12 * e12 * cos_lat;
            row[0] = dx; row[1] = dy; row[2] = dz;                                /* AddMsrtoDesign_BCEKMSVZ (ADJ:4710) */
            row[3] = -dx; row[4] = -dy; row[5] = -dz;
            break;
        }
        case 'C': case 'E': case 'M': {
            const double lat2 = a->geo[3 * (size_t)g2];
            if (type == 'E')
                a->t_val[t] = 2.0 * radius_in_chord_direction(X1, X2, lat1, lon1, lat2) *
                              sin(a->t_pre[t] / 2.0 / radius_in_chord_direction(X1, X2, lat1, lon1, lat2));
            if (type == 'M') a->t_val[t] = msl_arc_to_ellipsoid_chord(a->t_pre[t], lat1, lat2, net->stn_geoid[g1], net->stn_geoid[g2]);
            if (type != 'C') a->t_corr[t] = a->t_val[t] - a->t_pre[t];
            double d[3];
            *comp = chord_distance(X1, X2, lat1, lat2, a->geo[3 * (size_t)g1 + 2], a->geo[3 * (size_t)g2 + 2], d);
            for (int c = 0; c < 3; ++c) {
                row[c] = -d[c] / (*comp);
                row[3 + c] = d[c] / (*comp);
            }
            break;
        }
        case 'S': {
            double di[3], dt[3], d[3];
            height_offset(net->t_ih[t], lat1, lon1, di);
            height_offset(net->t_th[t], lat1, lon1, dt);                          /* (sic: station 1's position, ADJ:5466) */
            for (int c = 0; c < 3; ++c) d[c] = X2[c] - X1[c] + dt[c] - di[c];
            *comp = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (int c = 0; c < 3; ++c) {
                row[c] = -d[c] / (*comp);
                row[3 + c] = d[c] / (*comp);
            }
            break;
        }
        case 'V': case 'Z': {
            const double lat2 = a->geo[3 * (size_t)g2], lon2 = a->geo[3 * (size_t)g2 + 1];
            double e, n, up, dx, dy, dz;
            double zen = zenith_distance(X1, X2, lat1, lon1, lat2, lon2, net->t_ih[t], net->t_th[t], &e, &n, &up);
            double e2n2 = e * e + n * n, se = sqrt(e2n2);
            if (type == 'V') {
                *comp = zen;
                double se2n2_up2 = se / (up * up), up_se2n2 = up * se, cos2v = cos(*comp) * cos(*comp);
                dx = cos2v * (((e * sin_long + n * sin_lat * cos_long) / up_se2n2) + cos_lat * cos_long * se2n2_up2);
                dy = cos2v * (((-e * cos_long + n * sin_lat * sin_long) / up_se2n2) + cos_lat * sin_long * se2n2_up2);
                dz = cos2v * ((-n * cos_lat / up_se2n2) + sin_lat * se2n2_up2);
            } else {
                *comp = atan2(up, se);                                            /* VerticalAngle, :777 */
                double se_d = se / e2n2, up_d = up / (se * e2n2), cos2v = cos(*comp) * cos(*comp);
                dx = cos2v * ((-cos_lat * cos_long * se_d) - ((e * sin_long + n * sin_lat * cos_long) * up_d));
                dy = cos2v * ((-cos_lat * sin_long * se_d) + ((e * cos_long - n * sin_lat * sin_long) * up_d));
                dz = cos2v * ((-sin_lat * se_d) + (n * cos_lat * up_d));
            }
            row[0] = dx; row[1] = dy; row[2] = dz;
            row[3] = -dx; row[4] = -dy; row[5] = -dz;
            break;
        }
        case 'L': {
            const double lat2 = a->geo[3 * (size_t)g2];
            double nu1, nu2, Zn1, Zn2;
            double h2 = ellipsoid_height(X2, lat2, &nu2, &Zn2);
            double h1 = ellipsoid_height(X1, lat1, &nu1, &Zn1);
            *comp = h2 - h1;
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
            *comp = ellipsoid_height(X1, lat1, &nu1, &Zn1);
            row[0] = X1[0] / (nu1 + *comp);
            row[1] = X1[1] / (nu1 + *comp);
            row[2] = (X1[2] + Zn1) / (nu1 + *comp);
            break;
        }
        case 'I': case 'P': {
            /* UpdateDesignNormalMeasMatrices_IP (ADJ:5861): latitude of the estimates (CartToLat), row by "mechanical
             * differentiation": (lat(x + 1e-4 on one element) - lat) / 1e-4 (dnatemplategeodesyfuncs.hpp:282-320) */
            double lo, hh, lat0;
            cart_to_geo(X1, &lat0, &lo, &hh);
            for (int e = 0; e < 3; ++e) {
                double Y[3] = {X1[0], X1[1], X1[2]}, la;
                Y[e] += 1.0e-4;
                cart_to_geo(Y, &la, &lo, &hh);
                row[e] = (la - lat0) / 1.0e-4;
            }
            *comp = lat0;
            break;
        }
        case 'J': case 'Q': {
            /* UpdateDesignNormalMeasMatrices_JQ (ADJ:5931): computed = the station record's longitude;
             * term = x y / (x^2 + y^2)^1.5, d/dX = -term / cos(lon), d/dY = term / sin(lon), d/dZ = 0 */
            const double x = X1[0], y = X1[1];
            const double term = x * y / pow(x * x + y * y, 1.5);
            *comp = lon1;
            row[0] = term * -1. / cos(lon1);
            row[1] = term / sin(lon1);
            row[2] = 0.0;
            break;
        }
        default: *comp = 0.0;
    }
    (void)X3;
}

/* the one-time reductions applied when the matrices are first built (buildnewMatrices && !rebuildingDesign_):
 * deflection of the vertical (A ADJ:4790-4845, K ADJ:4940-4970, V ADJ:5523-5547, Z ADJ:5632-5656) and geoid
 * separation (L ADJ:5746-5753, H ADJ:5977-5984) */
static void tm_reduce(orc_adjustment* a, uint32_t t, const double* X1, const double* X2, const double* X3) {
    const orc_network* net = &a->net;
    const char type = net->t_type[t];
    const uint32_t g1 = net->t_stn[3 * (size_t)t], g2 = net->t_stn[3 * (size_t)t + 1], g3 = net->t_stn[3 * (size_t)t + 2];
    const double lat1 = a->geo[3 * (size_t)g1], lon1 = a->geo[3 * (size_t)g1 + 1];
    const double dV = net->stn_defl[2 * (size_t)g1], dM = net->stn_defl[2 * (size_t)g1 + 1];
    const int defl = fabs(dV) > ORC_E4_SEC_DEFLECTION || fabs(dM) > ORC_E4_SEC_DEFLECTION;
    double e, n, up;
    a->t_pre[t] = a->t_val[t];                                                    /* InitialiseMeasurement (ADJ:3928) */
    a->t_corr[t] = 0.0;
    switch (type) {
        case 'A': case 'D':
            if (defl) {
                double e12, n12, e13, n13;
                double d12 = direction(X1, X2, lat1, lon1, &e12, &n12), d13 = direction(X1, X3, lat1, lon1, &e13, &n13);
                if (d12 > d13) d13 += ORC_TWO_PI;
                double z12 = zenith_distance(X1, X2, lat1, lon1, a->geo[3 * (size_t)g2], a->geo[3 * (size_t)g2 + 1], net->t_ih[t], net->t_th[t], &e, &n, &up);
                double z13 = zenith_distance(X1, X3, lat1, lon1, a->geo[3 * (size_t)g3], a->geo[3 * (size_t)g3 + 1], net->t_ih[t], net->t_th[t], &e, &n, &up);
                /* HzAngleDeflectionCorrection (:1202) */
                a->t_corr[t] = (dM * sin(d13) - dV * cos(d13)) / tan(z13) - (dM * sin(d12) - dV * cos(d12)) / tan(z12);
                a->t_val[t] -= a->t_corr[t];
            }
            break;
        case 'K':
            if (defl) {
                double az = direction(X1, X2, lat1, lon1, &e, &n);
                double zen = zenith_distance(X1, X2, lat1, lon1, a->geo[3 * (size_t)g2], a->geo[3 * (size_t)g2 + 1], net->t_ih[t], net->t_th[t], &e, &n, &up);
                a->t_corr[t] = dV * tan(lat1) + ((dM * sin(az) - dV * cos(az)) / tan(zen));   /* LaplaceCorrection (:1181) */
                a->t_val[t] -= a->t_corr[t];
            }
            break;
        case 'V': case 'Z':
            if (defl) {
                double az = direction(X1, X2, lat1, lon1, &e, &n);
                a->t_corr[t] = dM * cos(az) + dV * sin(az);                        /* ZenithDeflectionCorrection (:1189) */
                if (type == 'V') a->t_val[t] += a->t_corr[t];
                else a->t_val[t] -= a->t_corr[t];
            }
            break;
        case 'L':
            if (fabs(net->stn_geoid[g1]) > 1.0e-4 || fabs(net->stn_geoid[g2]) > 1.0e-4) {
                a->t_corr[t] = net->stn_geoid[g2] - net->stn_geoid[g1];
                a->t_val[t] += a->t_corr[t];
            }
            break;
        case 'H':
            if (fabs(net->stn_geoid[g1]) > 1.0e-4) {
                a->t_corr[t] = net->stn_geoid[g1];
                a->t_val[t] += a->t_corr[t];
            }
            break;
        case 'I':   /* ADJ:5797-5804: deflection in the prime meridian */
            if (fabs(dM) > ORC_E4_SEC_DEFLECTION) {
                a->t_corr[t] = dM;
                a->t_val[t] -= a->t_corr[t];
            }
            break;
        case 'J':   /* ADJ:5828-5835: deflection in the prime vertical times sec(latitude) */
            if (fabs(dV) > ORC_E4_SEC_DEFLECTION) {
                a->t_corr[t] = dV / cos(lat1);
                a->t_val[t] -= a->t_corr[t];
            }
            break;
        default: break;
    }
}

/* direction sets: set of terrestrial measurement t (type 'D'), offset of its k x k weight matrix */
static uint32_t dset_of(const orc_adjustment* a, uint32_t t) {
    uint32_t s = 0;
    while (s + 1 < a->net.n_dsets && !(t >= a->net.dset_first[s] && t < a->net.dset_first[s] + a->net.dset_size[s])) ++s;
    return s;
}
static size_t dset_woff(const orc_adjustment* a, uint32_t s) {
    size_t o = 0;
    for (uint32_t q = 0; q < s; ++q) o += (size_t)a->net.dset_size[q] * a->net.dset_size[q];
    return o;
}

/* cml entry -> terrestrial measurement index, or -1 for a GNSS cluster */
static inline int64_t tm_index(const orc_adjustment* a, uint32_t entry) { return entry >= a->n_clusters ? (int64_t)entry - a->n_clusters : -1; }
static inline uint32_t entry_rows(const orc_adjustment* a, uint32_t entry) {
    return entry >= a->n_clusters ? 1u : 3u * (a->cl_off[entry + 1] - a->cl_off[entry]);
}
/* block-local coordinates of the measurement's stations */
static void tm_block_coords(const orc_adjustment* a, const blk_t* B, uint32_t t, const double* est, uint32_t* loc, const double** X) {
    static const double zero3[3] = {0, 0, 0};
    int ns = tm_station_count(a->net.t_type[t]);
    for (int s = 0; s < 3; ++s) {
        if (s < ns) {
            loc[s] = local_index(B, a->net.t_stn[3 * (size_t)t + s]);
            X[s] = est + 3 * (size_t)loc[s];
        } else {
            loc[s] = 0;
            X[s] = zero3;
        }
    }
}

/* UpdateNormals (ADJ:1364): GNSS measurements, and UpdateNormals_A / _BCEKLMSVZ / _HIJPQR (ADJ:1524-1662) with the
 * design rows stored by compute_b */
static void update_normals(orc_adjustment* a, blk_t* B) {
    uint32_t trow = 0;
    for (uint32_t c = 0; c < B->n_cml; ++c) {
        uint32_t cl = B->cml[c];
        int64_t t = tm_index(a, cl);
        if (t >= 0 && a->net.t_type[t] == 'D') {
            /* UpdateAtVinv_D (ADJ:1328) + UpdateNormals_D (ADJ:1540): A^T W A over the whole set, W dense (LoadVarianceMatrix_D);
             * the k angles of a set are consecutive entries, the set is added when its first angle comes up */
            const uint32_t s0 = dset_of(a, (uint32_t)t);
            if (a->net.dset_first[s0] == (uint32_t)t) {
                const uint32_t k = a->net.dset_size[s0];
                const double* Wd = a->net.dset_w + dset_woff(a, s0);
                for (uint32_t x = 0; x < k; ++x)
                    for (uint32_t y = 0; y < k; ++y) {
                        const double w = Wd[x + (size_t)y * k];
                        const double* rx = B->trow + 9 * (size_t)(trow + x);
                        const double* ry = B->trow + 9 * (size_t)(trow + y);
                        for (int p = 0; p < 3; ++p)
                            for (int q = 0; q < 3; ++q) {
                                const uint32_t lp = 3 * local_index(B, a->net.t_stn[3 * (size_t)(t + x) + p]);
                                const uint32_t lq = 3 * local_index(B, a->net.t_stn[3 * (size_t)(t + y) + q]);
                                for (int cc = 0; cc < 3; ++cc)
                                    for (int r = 0; r < 3; ++r)
                                        if (lp + r >= lq + cc) lower_add(B->N, B->n, lp + r, lq + cc, (w * rx[3 * p + r]) * ry[3 * q + cc]);
                            }
                    }
            }
            trow++;
            continue;
        }
        if (t >= 0) {
            const double* row = B->trow + 9 * (size_t)trow++;
            const double w = 1.0 / a->net.t_var[t];                               /* UpdateAtVinv (ADJ:1288) */
            const int ns = tm_station_count(a->net.t_type[t]);
            uint32_t loc[3];
            for (int q = 0; q < ns; ++q) loc[q] = 3 * local_index(B, a->net.t_stn[3 * (size_t)t + q]);
            for (int p = 0; p < ns; ++p)
                for (int q = 0; q < ns; ++q)
                    for (int cc = 0; cc < 3; ++cc)
                        for (int r = 0; r < 3; ++r) lower_add(B->N, B->n, loc[p] + r, loc[q] + cc, (w * row[3 * p + r]) * row[3 * q + cc]);
            continue;
        }
        uint32_t i = a->cl_off[cl];
        if (!a->cl_W[cl]) {
            uint32_t s1 = 3 * local_index(B, a->net.stn1[i]);
            uint32_t s2 = 3 * local_index(B, a->net.stn2[i]);
            update_normals_G(B->N, B->n, s1, s2, a->W + (size_t)i * 6);
        } else
            cluster_contribution(a, B, cl, B->N, NULL, NULL);
    }
}

/* FillDesignNormalMeasurementsMatrices(false) (ADJ:3888) -> UpdateDesignMeasMatrices_GX (ADJ:5283):
 * b = term1 - (x2 - x1), AddMsrtoMeasMinusComp (ADJ:4719) */
static void compute_b(orc_adjustment* a, blk_t* B, const double* est) {
    uint32_t row = 0, trow = 0;
    for (uint32_t c = 0; c < B->n_cml; ++c) {
        uint32_t cl = B->cml[c];
        int64_t t = tm_index(a, cl);
        if (t >= 0) {
            /* the type's UpdateDesignNormalMeasMatrices_* + AddMsrtoMeasMinusComp (ADJ:4719) */
            uint32_t loc[3];
            const double* X[3];
            double comp;
            tm_block_coords(a, B, (uint32_t)t, est, loc, X);
            tm_evaluate(a, (uint32_t)t, X[0], X[1], X[2], &comp, B->trow + 9 * (size_t)trow++);
            double mmc = a->t_val[t] - comp;
            switch (a->net.t_type[t]) {
                case 'A': case 'B': case 'D': case 'K':
                    if (mmc < -5.5) mmc += ORC_TWO_PI;
                    else if (mmc > 5.5) mmc -= ORC_TWO_PI;
                default: break;
            }
            B->b[row++] = mmc;
            continue;
        }
        for (uint32_t i = a->cl_off[cl]; i < a->cl_off[cl + 1]; ++i, row += 3) {
            uint32_t s2 = 3 * local_index(B, a->net.stn2[i]);
            if (a->net.stn1[i] == 0xffffffffu) {
                /* point cluster: computed = station coordinate (ADJ:6343) */
                for (int k = 0; k < 3; ++k) B->b[row + k] = a->net.obs[3 * (size_t)i + k] - est[s2 + k];
            } else {
                uint32_t s1 = 3 * local_index(B, a->net.stn1[i]);
                for (int k = 0; k < 3; ++k) B->b[row + k] = a->net.obs[3 * (size_t)i + k] - (est[s2 + k] - est[s1 + k]);
            }
        }
    }
}

/* UpdateGeographicCoords (ADJ:8734) / UpdateGeographicCoordsPhased (ADJ:8711): the station records take the geodetic
 * coordinates of the block's estimates (first appearance only); CartToGeo = dnatemplategeodesyfuncs.hpp:154-225 */
static void cart_to_geo(const double* X, double* lat, double* lon, double* h) {
    double a_, e2;
    grs80(&a_, &e2);
    const double b_ = a_ * (1.0 - 1.0 / 298.257222101);
    double x = X[0], y = X[1], z = X[2];
    double p2 = x * x + y * y, p = sqrt(p2), a2 = a_ * a_, b2 = b_ * b_, Z2 = z * z;
    double a2Z2 = a2 * Z2, b2p2 = b2 * p2, A = a2Z2 + b2p2;
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
    double p_E = a2 * p / (a2 + twom), Z_E = b2 * z / (b2 + twom);
    *lat = atan(a2 * Z_E / (b2 * p_E));
    *lon = atan(y / x);
    if (x < 0.0 && y > 0.0) *lon += ORC_PI;
    else if (x < 0.0 && y < 0.0) *lon = -(ORC_PI - *lon);
    *h = sqrt(((p - p_E) * (p - p_E)) + ((z - Z_E) * (z - Z_E)));
    if ((p + fabs(z)) < (p_E + fabs(Z_E))) *h *= -1.;
}
static void update_geographic(orc_adjustment* a, const blk_t* B, const double* est) {
    if (!a->n_tm) return;
    for (uint32_t p = 0; p < B->n_stn; ++p) {
        if (!B->first_fwd[p]) continue;
        double* g = a->geo + 3 * (size_t)B->stations[p];
        cart_to_geo(est + 3 * (size_t)p, &g[0], &g[1], &g[2]);
    }
}

/* FormConstraintStationVarianceMatrix (ADJ:2041-2137): the 3x3 weight matrix of a station's constraint.  CCC / FFF:
 * 1/var on the diagonal.  Mixed codes: variances per axis of the LOCAL frame -- for geographic station records the first
 * character is the latitude (north), the second the longitude (east); otherwise first = east / X, second = north / Y; third =
 * up -- propagated to cartesian with the station's current latitude / longitude (unless the record is cartesian), then inverted. */
static int constraint_matrix(const orc_adjustment* a, uint32_t stn, double W[3][3]) {
    const orc_network* net = &a->net;
    const char* c = net->constraints + 3 * (size_t)stn;
    memset(W, 0, 9 * sizeof(double));
    if (c[0] == 'C' && c[1] == 'C' && c[2] == 'C') {
        W[0][0] = W[1][1] = W[2][2] = 1. / a->var_C;
        return 0;
    }
    if (c[0] == 'F' && c[1] == 'F' && c[2] == 'F') {
        W[0][0] = W[1][1] = W[2][2] = 1. / a->var_F;
        return 0;
    }
    if (!net->stn_type || !net->stn_llh) return -1;
    const int type = net->stn_type[stn];          /* 0 XYZ, 1 LLh, 2 LLH, 3 UTM (dnatypes-structs.hpp) */
    const int geographic = type == 1 || type == 2;
    const double v0 = c[0] == 'F' ? a->var_F : a->var_C, v1 = c[1] == 'F' ? a->var_F : a->var_C, v2 = c[2] == 'F' ? a->var_F : a->var_C;
    double vl[3];
    vl[geographic ? 1 : 0] = v0;
    vl[geographic ? 0 : 1] = v1;
    vl[2] = v2;
    double V[6];                                   /* packed lower, order 3 */
    if (type == 0) {
        V[0] = vl[0]; V[1] = 0; V[2] = 0; V[3] = vl[1]; V[4] = 0; V[5] = vl[2];
    } else {
        const double* g = a->geo ? a->geo + 3 * (size_t)stn : net->stn_llh + 3 * (size_t)stn;
        const double sl = sin(g[0]), cl = cos(g[0]), so = sin(g[1]), co = cos(g[1]);
        /* columns: east, north, up in cartesian components (PropagateVariances_LocalCart, local -> cart) */
        const double R[3][3] = {{-so, -sl * co, cl * co}, {co, -sl * so, cl * so}, {0.0, cl, sl}};
        for (int col = 0; col < 3; ++col)
            for (int r = col; r < 3; ++r) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += R[r][k] * vl[k] * R[col][k];
                V[orc_packed_index(3, r, col)] = s;
            }
    }
    if (orc_inverse_normals_packed(V, 3, 0)) return -2;      /* FormInverseVarianceMatrix */
    for (int col = 0; col < 3; ++col)
        for (int r = col; r < 3; ++r) W[r][col] = W[col][r] = V[orc_packed_index(3, r, col)];
    return 0;
}

/* blockadd / blocksubtract of the 3x3 constraint weight, packed dest */
static void add_constraint(double* N, uint32_t n, uint32_t s, double W[3][3], double sign) {
    for (int col = 0; col < 3; ++col)
        for (int r = col; r < 3; ++r) N[orc_packed_index(n, s + r, s + col)] += sign * W[r][col];
}

/* AddConstraintStationstoNormals{Forward ADJ:1884, Reverse :1923, Combine :1960, Simultaneous :2010} */
enum { CON_FWD, CON_REV, CON_CMB, CON_SIM };
static int add_constraints(orc_adjustment* a, blk_t* B, int which) {
    for (uint32_t p = 0; p < B->n_stn; ++p) {
        double W[3][3], sign = 1.0;
        if (which == CON_FWD && !B->first_fwd[p]) continue;
        if (which == CON_REV && !B->first_rev[p]) continue;
        if (which == CON_CMB) {
            if (B->first_fwd[p]) continue;
            sign = -1.0;
        }
        if (constraint_matrix(a, B->stations[p], W)) {
            snprintf(a->err, sizeof(a->err), "oracle: the constraint of station %u needs the station type and position (stn_type, stn_llh)", B->stations[p]);
            return -1;
        }
        add_constraint(B->N, B->n, 3 * p, W, sign);
    }
    return 0;
}

/* Solve (ADJ:6586-6667) */
static int solve(orc_adjustment* a, blk_t* B, int compute_inverse, uint32_t block) {
    uint32_t n = B->n;
    if (compute_inverse) {
        int info = orc_inverse_normals_packed(B->N, n, a->set.scale_normals_to_unity);
        a->solves++;
        a->sum_n3 += (double)n * n * n;
        if (info || isnan(B->N[0]) || isinf(B->N[0])) {
            snprintf(a->err, sizeof(a->err), "Matrix inversion failed, the matrix is singular. (block %u, info %d)", block + 1, info);
            return -1;
        }
    }
    /* At_Vinv_m = AtVinv * measMinusComp (ADJ:6659-6660); AtVinv is never materialised */
    const double trhs0 = orc_now();
    double* rhs = (double*)calloc(n ? n : 1, sizeof(double));
    uint32_t brow = 0;
    uint32_t trow = 0;
    for (uint32_t c = 0; c < B->n_cml; ++c) {
        uint32_t cl = B->cml[c];
        int64_t t = tm_index(a, cl);
        if (t >= 0 && a->net.t_type[t] == 'D') {
            /* At_Vinv_m of a direction set: a_x^T sum_y W_xy b_y */
            const uint32_t s0 = dset_of(a, (uint32_t)t);
            const uint32_t first = a->net.dset_first[s0], k = a->net.dset_size[s0], x = (uint32_t)t - first;
            const double* Wd = a->net.dset_w + dset_woff(a, s0);
            double wb = 0.0;
            for (uint32_t y = 0; y < k; ++y) wb += Wd[x + (size_t)y * k] * B->b[brow - x + y];
            const double* row = B->trow + 9 * (size_t)trow++;
            brow++;
            for (int q = 0; q < 3; ++q) {
                uint32_t l = 3 * local_index(B, a->net.t_stn[3 * (size_t)t + q]);
                for (int r = 0; r < 3; ++r) rhs[l + r] += row[3 * q + r] * wb;
            }
            continue;
        }
        if (t >= 0) {
            const double* row = B->trow + 9 * (size_t)trow++;
            const double wb = (1.0 / a->net.t_var[t]) * B->b[brow++];
            const int ns = tm_station_count(a->net.t_type[t]);
            for (int q = 0; q < ns; ++q) {
                uint32_t l = 3 * local_index(B, a->net.t_stn[3 * (size_t)t + q]);
                for (int r = 0; r < 3; ++r) rhs[l + r] += row[3 * q + r] * wb;
            }
            continue;
        }
        uint32_t i = a->cl_off[cl];
        const double* bb = B->b + brow;
        brow += 3 * (a->cl_off[cl + 1] - i);
        if (a->cl_W[cl]) {
            cluster_contribution(a, B, cl, NULL, rhs, bb);
            continue;
        }
        const double* w = a->W + (size_t)i * 6;
        uint32_t s1 = 3 * local_index(B, a->net.stn1[i]);
        uint32_t s2 = 3 * local_index(B, a->net.stn2[i]);
        for (int r = 0; r < 3; ++r) {
            double wb = (w[sym6(r, 0)] * bb[0] + w[sym6(r, 1)] * bb[1]) + w[sym6(r, 2)] * bb[2];
            rhs[s1 + r] += -wb;
            rhs[s2 + r] += wb;
        }
    }
    for (uint32_t q = 0; q < B->n_ps; ++q) {
        const pseudo_set* P = &B->ps[q];
        uint32_t nj = 3 * P->k;
        for (uint32_t i = 0; i < nj; ++i) {
            double acc = 0.0;
            for (uint32_t j = 0; j < nj; ++j) acc += packed_get(P->W, nj, i, j) * B->b[P->row0 + j];
            rhs[3 * local_index(B, P->stn[i / 3]) + i % 3] += acc;
        }
    }
    /* corrections = N^-1 * At_Vinv_m (multiply_sym, ADJ:6665) */
    double ts0 = orc_now();
    orc_t[T_RHS] += ts0 - trhs0;
    orc_multiply_sym_packed(B->N, rhs, B->corr, n);
    orc_t[T_SYMV] += orc_now() - ts0;
    free(rhs);
    return 0;
}

/* matrix_2d::compute_maximum_value (MAT:1532) on a column vector */
static double max_value(const double* v, uint32_t n) {
    uint32_t best = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (fabs(v[i]) > fabs(v[best])) best = i;
    return n ? v[best] : 0.0;
}

static void free_pseudo(blk_t* B) {
    for (uint32_t q = 0; q < 2; ++q) {
        free(B->ps[q].W);
        B->ps[q].W = NULL;
        free(B->ps[q].stn);
        B->ps[q].stn = NULL;
    }
    B->n_ps = 0;
}

/* shrink(): drop the last `sets` pseudo-measurement sets (rows of b / columns of AtVinv) */
static void shrink_pseudo(blk_t* B, uint32_t sets) {
    while (sets-- && B->n_ps) {
        pseudo_set* P = &B->ps[B->n_ps - 1];
        B->b_rows -= 3 * P->k;
        free(P->W);
        P->W = NULL;
        free(P->stn);
        P->stn = NULL;
        B->n_ps--;
    }
}

/* Steps 1-2 of CarryStnEstimatesandVariances{Forward ADJ:1006-1048, Reverse ADJ:1160-1202}:
 * gather the junction block of the a-posteriori variances and the junction estimates, invert. */
static int gather_junctions(orc_adjustment* a, const blk_t* From, const double* apost, const double* est, const uint32_t* jsl,
                            uint32_t k, double* jvar, double* jest) {
    uint32_t nj = 3 * k;
    for (uint32_t p = 0; p < k; ++p) {
        uint32_t sp = 3 * local_index(From, jsl[p]);
        for (int c = 0; c < 3; ++c) jest[3 * p + c] = est[sp + c];
        for (uint32_t q = p; q < k; ++q) {
            uint32_t sq = 3 * local_index(From, jsl[q]);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    uint32_t i = 3 * q + r, j = 3 * p + c; /* lower part of the junction matrix */
                    if (i < j) continue;
                    jvar[orc_packed_index(nj, i, j)] = packed_get(apost, From->n, sq + r, sp + c);
                }
        }
    }
    if (nj == 0) return 0;
    int info = orc_inverse_normals_packed(jvar, nj, 0); /* FormInverseVarianceMatrix(junctionVariances) */
    a->solves += 0;
    if (info) {
        snprintf(a->err, sizeof(a->err), "Matrix inversion failed, the matrix is singular. (junction variances, info %d)", info);
        return -1;
    }
    return 0;
}

/* Steps 4-5 of the carry functions (ADJ:1053-1127, 1204-1280, 3249-3319): grow b / AtVinv of
 * the destination block by the junction pseudo measurements, add W_J into its normals. */
static void attach_junctions(blk_t* To, const uint32_t* jsl, uint32_t k, const double* jvar, const double* jest) {
    uint32_t nj = 3 * k;
    pseudo_set* P = &To->ps[To->n_ps++];
    P->k = k;
    P->row0 = To->b_rows;
    P->stn = (uint32_t*)malloc((k ? k : 1) * sizeof(uint32_t));
    memcpy(P->stn, jsl, k * sizeof(uint32_t));
    P->W = (double*)malloc((psize(nj) ? psize(nj) : 1) * sizeof(double));
    memcpy(P->W, jvar, psize(nj) * sizeof(double));
    To->b_rows += nj;
    for (uint32_t p = 0; p < k; ++p) {
        uint32_t sp = 3 * local_index(To, jsl[p]);
        for (int c = 0; c < 3; ++c) To->b[P->row0 + 3 * p + c] = jest[3 * p + c] - To->est[sp + c];
        for (uint32_t q = 0; q < k; ++q) {
            uint32_t sq = 3 * local_index(To, jsl[q]);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) lower_add(To->N, To->n, sq + r, sp + c, packed_get(jvar, nj, 3 * q + r, 3 * p + c));
        }
    }
}

static void update_max_corr(orc_adjustment* a, const double* corr, uint32_t n) {
    double mv = max_value(corr, n);
    if (fabs(mv) > fabs(a->maxCorr)) a->maxCorr = mv;
}

orc_adjustment* orc_adjust_create(const orc_network* net, const orc_settings* set, int phased) {
    orc_adjustment* a = (orc_adjustment*)calloc(1, sizeof(*a));
    a->net = *net;
    a->set = *set;
    a->phased = phased;
    a->var_C = set->fixed_std_dev * set->fixed_std_dev; /* ADJ:232-245 */
    a->var_F = set->free_std_dev * set->free_std_dev;
    a->n_blocks = phased ? net->n_blocks : 1;
    a->blk = (blk_t*)calloc(a->n_blocks, sizeof(blk_t));
    if (set->threads > 0) {
        char buf[32];
        snprintf(buf, sizeof(buf), "%d", set->threads);
        setenv("MKL_NUM_THREADS", buf, 1);
        setenv("OMP_NUM_THREADS", buf, 1);
        if (ext_handle) {
            void (*setn)(int) = (void (*)(int))dlsym(ext_handle, "MKL_Set_Num_Threads");
            if (setn) setn(set->threads);
        }
    }
    return a;
}

void orc_adjust_destroy(orc_adjustment* a) {
    if (!a) return;
    for (uint32_t b = 0; b < a->n_blocks; ++b) {
        blk_t* B = &a->blk[b];
        free_pseudo(B);
        free(B->stations); free(B->first_fwd); free(B->first_rev);
        free(B->N); free(B->NR); free(B->est); free(B->orig); free(B->rig); free(B->rigvar);
        free(B->corr); free(B->corrR); free(B->b); free(B->jvar); free(B->jvarFwd); free(B->jestFwd); free(B->jestRev);
        free(B->prec);
        free(B->trow);
    }
    free(a->t_val); free(a->t_pre); free(a->t_corr); free(a->geo);
    free(a->osc_prev); free(a->osc_seen); free(a->osc_cnt); free(a->osc_slot); free(a->osc_hist);
    for (int f = 0; f < 8; ++f) free(a->tm_field[f]);
    for (int f = 0; f < 7; ++f) free(a->msr_field[f]);
    free(a->blk);
    free(a->W);
    if (a->cl_W)
        for (uint32_t c = 0; c < a->n_clusters; ++c) free(a->cl_W[c]);
    free(a->cl_W);
    free(a->cl_off);
    free(a->simul_stations);
    free(a->simul_cml);
    free(a);
}

/* PrepareAdjustment (ADJ:258) -> LoadSegmentationMetrics (ADJ:10426), CreateStnAppearanceList (SEG:432),
 * PrepareAdjustmentBlock (ADJ:2873) */
int orc_adjust_prepare(orc_adjustment* a) {
    const orc_network* net = &a->net;
    /* clusters: without cluster arrays every vector is a single 'G' baseline */
    a->n_clusters = net->n_clusters ? net->n_clusters : net->n_baselines;
    a->cl_off = (uint32_t*)malloc(((size_t)a->n_clusters + 1) * sizeof(uint32_t));
    a->cl_W = (double**)calloc((size_t)a->n_clusters + 1, sizeof(double*));
    for (uint32_t c = 0; c <= a->n_clusters; ++c) a->cl_off[c] = net->n_clusters ? net->cluster_off[c] : c;
    /* terrestrial measurements and the geodetic station data they need */
    a->n_tm = net->n_tmsr;
    if (a->n_tm) {
        a->t_val = (double*)malloc(a->n_tm * sizeof(double));
        a->t_pre = (double*)malloc(a->n_tm * sizeof(double));
        a->t_corr = (double*)calloc(a->n_tm, sizeof(double));
        memcpy(a->t_val, net->t_value, a->n_tm * sizeof(double));
        memcpy(a->t_pre, net->t_value, a->n_tm * sizeof(double));
        a->geo = (double*)malloc(3 * (size_t)net->n_stations * sizeof(double));
        memcpy(a->geo, net->stn_llh, 3 * (size_t)net->n_stations * sizeof(double));
    }
    /* measurement weights: LoadVarianceMatrix_G / _X / _Y (ADJ:4214 / 4312 / 4494) + FormInverseVarianceMatrix (ADJ:8472) */
    a->W = (double*)calloc((size_t)net->n_baselines * 6 + 1, sizeof(double));
    size_t voff = 0;
    for (uint32_t c = 0; c < a->n_clusters; ++c) {
        uint32_t i = a->cl_off[c], k = a->cl_off[c + 1] - i, nc = 3 * k;
        const double* V = net->n_clusters ? net->cluster_vcv + voff : NULL;
        voff += (size_t)nc * nc;
        if (k == 1 && net->stn1[i] != 0xffffffffu) {
            double six[6];
            const double* v6 = net->n_clusters ? six : net->vcv6 + (size_t)i * 6;
            if (net->n_clusters) {
                six[0] = V[0]; six[1] = V[3]; six[2] = V[4]; six[3] = V[6]; six[4] = V[7]; six[5] = V[8];
            }
            if (orc_weight_3x3(v6, a->W + (size_t)i * 6)) {
                snprintf(a->err, sizeof(a->err), "Matrix inversion failed, the matrix is singular. (variance matrix of measurement %u)", c);
                return -1;
            }
            continue;
        }
        double* Wc = (double*)malloc((size_t)nc * nc * sizeof(double));
        memcpy(Wc, V, (size_t)nc * nc * sizeof(double));
        if (orc_cholesky_inverse_full(Wc, nc, nc)) {
            free(Wc);
            snprintf(a->err, sizeof(a->err), "Matrix inversion failed, the matrix is singular. (variance matrix of measurement %u)", c);
            return -1;
        }
        a->cl_W[c] = Wc;
        for (uint32_t j = 0; j < k; ++j) {
            double* w = a->W + (size_t)(i + j) * 6;
            const double* D = Wc + (size_t)(3 * j) * nc + 3 * j;
            w[0] = D[0]; w[1] = D[nc]; w[2] = D[nc + 1]; w[3] = D[2 * nc]; w[4] = D[2 * nc + 1]; w[5] = D[2 * nc + 2];
        }
    }
    if (!a->phased) {
        a->simul_stations = (uint32_t*)malloc((net->n_stations + 1) * sizeof(uint32_t));
        for (uint32_t s = 0; s < net->n_stations; ++s) a->simul_stations[s] = s;
        a->simul_cml = (uint32_t*)malloc(((size_t)a->n_clusters + a->n_tm + 1) * sizeof(uint32_t));
        for (uint32_t i = 0; i < a->n_clusters + a->n_tm; ++i) a->simul_cml[i] = i;
        /* (an explicit order may come with the network: cml_off/cml of "block 0") */
        if (net->n_blocks == 1 && net->cml_off && net->cml_off[1] == a->n_clusters + a->n_tm)
            memcpy(a->simul_cml, net->cml, (a->n_clusters + a->n_tm) * sizeof(uint32_t));
    }
    uint32_t prev_net = 999999;
    for (uint32_t b = 0; b < a->n_blocks; ++b) {
        blk_t* B = &a->blk[b];
        if (a->phased) {
            uint32_t ni = net->isl_off[b + 1] - net->isl_off[b], nj = net->jsl_off[b + 1] - net->jsl_off[b];
            B->n_stn = ni + nj;
            B->stations = (uint32_t*)malloc((B->n_stn + 1) * sizeof(uint32_t));
            memcpy(B->stations, net->isl + net->isl_off[b], ni * sizeof(uint32_t));
            memcpy(B->stations + ni, net->jsl + net->jsl_off[b], nj * sizeof(uint32_t));
            qsort(B->stations, B->n_stn, sizeof(uint32_t), cmp_u32);
            B->n_jsl = nj;
            B->jsl = net->jsl + net->jsl_off[b];
            B->n_cml = net->cml_off[b + 1] - net->cml_off[b];
            B->cml = net->cml + net->cml_off[b];
            B->first = (net->net_id[b] != prev_net);
            prev_net = net->net_id[b];
            B->last = (b + 1 == a->n_blocks) || (net->net_id[b] != net->net_id[b + 1]);
            B->isolated = B->first && B->last;
        } else {
            B->n_stn = net->n_stations;
            B->stations = (uint32_t*)malloc((B->n_stn + 1) * sizeof(uint32_t));
            memcpy(B->stations, a->simul_stations, B->n_stn * sizeof(uint32_t));
            B->n_jsl = 0;
            B->n_cml = a->n_clusters + a->n_tm;
            B->cml = a->simul_cml;
            B->first = B->last = B->isolated = 1;
        }
        B->n = 3 * B->n_stn;
        B->m = 0;
        B->n_trow = 0;
        for (uint32_t q = 0; q < B->n_cml; ++q) {
            B->m += entry_rows(a, B->cml[q]);
            if (tm_index(a, B->cml[q]) >= 0) B->n_trow++;
        }
        B->trow = (double*)calloc(9 * (size_t)B->n_trow + 1, sizeof(double));
        B->first_fwd = (uint8_t*)calloc(B->n_stn + 1, 1);
        B->first_rev = (uint8_t*)calloc(B->n_stn + 1, 1);
    }
    /* CreateStnAppearanceList (SEG:432-487) */
    {
        uint8_t* seen = (uint8_t*)calloc(net->n_stations + 1, 1);
        for (uint32_t b = 0; b < a->n_blocks; ++b)
            for (uint32_t p = 0; p < a->blk[b].n_stn; ++p)
                if (!seen[a->blk[b].stations[p]]) {
                    seen[a->blk[b].stations[p]] = 1;
                    a->blk[b].first_fwd[p] = 1;
                }
        memset(seen, 0, net->n_stations + 1);
        for (uint32_t bb = a->n_blocks; bb-- > 0;)
            for (uint32_t p = 0; p < a->blk[bb].n_stn; ++p)
                if (!seen[a->blk[bb].stations[p]]) {
                    seen[a->blk[bb].stations[p]] = 1;
                    a->blk[bb].first_rev[p] = 1;
                }
        free(seen);
    }
    for (uint32_t b = 0; b < a->n_blocks; ++b) {
        blk_t* B = &a->blk[b];
        uint32_t n = B->n;
        /* pseudo measurement capacity (ADJ:822-834): JSL(b) + JSL(b-1) */
        uint32_t pseudo = 0;
        if (a->phased && !B->isolated) {
            pseudo = B->n_jsl;
            if (!B->first) pseudo += a->blk[b - 1].n_jsl;
        }
        B->N = (double*)calloc(psize(n) + 1, sizeof(double));
        B->NR = (double*)calloc(psize(n) + 1, sizeof(double));
        B->rigvar = (double*)calloc(psize(n) + 1, sizeof(double));
        B->est = (double*)calloc(n + 1, sizeof(double));
        B->orig = (double*)calloc(n + 1, sizeof(double));
        B->rig = (double*)calloc(n + 1, sizeof(double));
        B->corr = (double*)calloc(n + 1, sizeof(double));
        B->corrR = (double*)calloc(n + 1, sizeof(double));
        B->b = (double*)calloc(B->m + 3 * pseudo + 1, sizeof(double));
        B->b_rows = B->m;
        uint32_t nj = 3 * B->n_jsl;
        B->jvar = (double*)calloc(psize(nj) + 1, sizeof(double));
        B->jvarFwd = (double*)calloc(psize(nj) + 1, sizeof(double));
        B->jestFwd = (double*)calloc(nj + 1, sizeof(double));
        B->jestRev = (double*)calloc((b > 0 && a->phased ? 3 * a->blk[b - 1].n_jsl : 0) + 1, sizeof(double));
        /* PopulateEstimatedStationMatrix (ADJ:632) */
        for (uint32_t p = 0; p < B->n_stn; ++p)
            for (int c = 0; c < 3; ++c) B->est[3 * p + c] = B->orig[3 * p + c] = B->rig[3 * p + c] = net->xyz0[3 * (size_t)B->stations[p] + c];
        /* FillDesignNormalMeasurementsMatrices(true) (ADJ:913): one-time reductions of the terrestrial measurements, b,
         * AtVinv, N */
        for (uint32_t q = 0; q < B->n_cml; ++q) {
            int64_t t = tm_index(a, B->cml[q]);
            if (t < 0) continue;
            uint32_t loc[3];
            const double* X[3];
            tm_block_coords(a, B, (uint32_t)t, B->est, loc, X);
            tm_reduce(a, (uint32_t)t, X[0], X[1], X[2]);
        }
        compute_b(a, B, B->est);
        update_normals(a, B);
        /* back up (ADJ:2955), then constraints (ADJ:2961-2970) */
        memcpy(B->NR, B->N, psize(n) * sizeof(double));
        if (add_constraints(a, B, a->phased ? CON_FWD : CON_SIM)) return -1;
    }
    return 0;
}

/* AdjustSimultaneous (ADJ:2413-2511) for a GNSS-only network */
/* dna_adjust::UpdateIterationDiagnostics (ADJ:7450-7554): block by block, station by station, the correction of this iteration against the
 * one the station was last seen with; two anti-parallel turns of similar size in a row make a record.  The reference keeps the last
 * correction of a record in the station's local frame (Rotate_CartLocal at the .bst position); the magnitudes are those of the same
 * vector, so the record here holds the cartesian correction and the tests rotate. */
static int update_iteration_diagnostics(orc_adjustment* a) {
    const uint32_t ns = a->net.n_stations;
    if (!a->osc_prev) {
        a->osc_prev = (double*)calloc(3 * (size_t)ns + 1, sizeof(double));
        a->osc_seen = (uint8_t*)calloc((size_t)ns + 1, 1);
        a->osc_cnt = (uint32_t*)calloc((size_t)ns + 1, sizeof(uint32_t));
        a->osc_slot = (int32_t*)malloc(((size_t)ns + 1) * sizeof(int32_t));
        a->osc_hist = (orc_osc_record*)calloc((size_t)ns + 1, sizeof(orc_osc_record));
        if (!a->osc_prev || !a->osc_seen || !a->osc_cnt || !a->osc_slot || !a->osc_hist) return -1;
        for (uint32_t s = 0; s < ns; ++s) a->osc_slot[s] = -1;
    }
    for (uint32_t b = 0; b < a->n_blocks; ++b) {
        const blk_t* B = &a->blk[b];
        for (uint32_t s = 0; s < B->n_stn; ++s) {
            const uint32_t g = B->stations[s];
            const double cx = B->corr[3 * s], cy = B->corr[3 * s + 1], cz = B->corr[3 * s + 2];
            const double magCurr = sqrt(cx * cx + cy * cy + cz * cz);
            double* p = &a->osc_prev[3 * (size_t)g];
            if (!a->osc_seen[g]) {                                   /* first time seeing this station: store and move on */
                a->osc_seen[g] = 1;
                p[0] = cx; p[1] = cy; p[2] = cz;
                continue;
            }
            const double px = p[0], py = p[1], pz = p[2];
            const double magPrev = sqrt(px * px + py * py + pz * pz);
            p[0] = cx; p[1] = cy; p[2] = cz;
            if (magCurr < 0.001 && magPrev < 0.001) {                /* sub-millimetre */
                a->osc_cnt[g] = 0;
                continue;
            }
            const double dot = cx * px + cy * py + cz * pz, denom = magCurr * magPrev;
            const double cosAngle = denom > 1e-30 ? dot / denom : 0.0;
            const double ratio = magPrev > 1e-30 ? magCurr / magPrev : 0.0;
            if (cosAngle < -0.5 && ratio > 0.3 && ratio < 3.0)
                a->osc_cnt[g]++;
            else
                a->osc_cnt[g] = 0;
            if (a->osc_cnt[g] >= 2) {
                orc_osc_record* r;
                if (a->osc_slot[g] < 0) {
                    a->osc_slot[g] = (int32_t)a->n_osc;
                    r = &a->osc_hist[a->n_osc++];
                    r->station = g;
                    r->first_iteration = a->iterations;
                    r->first_mag = magCurr;
                } else {
                    r = &a->osc_hist[a->osc_slot[g]];
                }
                r->last_iteration = a->iterations;
                r->cycles = a->osc_cnt[g];
                r->last_mag = magCurr;
                r->cx = cx; r->cy = cy; r->cz = cz;
            }
        }
    }
    return 0;
}

uint32_t orc_adjust_oscillation_history(const orc_adjustment* a, orc_osc_record* out, uint32_t cap) {
    for (uint32_t i = 0; out && i < a->n_osc && i < cap; ++i) out[i] = a->osc_hist[i];
    return a->n_osc;
}

static int adjust_simultaneous(orc_adjustment* a) {
    blk_t* B = &a->blk[0];
    a->iterations = 0;
    for (uint32_t i = 0; i < a->set.max_iterations; ++i) {
        a->iterations++;
        if (solve(a, B, a->iterations < 2 || a->n_tm, 0)) return ORC_ADJUST_EXCEPTION_RAISED;   /* ADJ:2457 */
        for (uint32_t k = 0; k < B->n; ++k) B->est[k] += B->corr[k];                      /* ADJ:2463 */
        a->maxCorr = max_value(B->corr, B->n);                                            /* ADJ:2466 */
        if (update_iteration_diagnostics(a)) return ORC_ADJUST_EXCEPTION_RAISED;          /* ADJ:2468 */
        if (a->iterations <= 64) a->max_corr_hist[a->iterations - 1] = a->maxCorr;
        if (!(fabs(a->maxCorr) > a->set.iteration_threshold)) break;                      /* ADJ:2477 */
        int last = (i + 1 >= a->set.max_iterations);
        /* UpdateAdjustment(!last) (ADJ:473): geodetic coordinates of the stations (non-GPS networks, ADJ:541-545), new
         * meas-minus-computed and design; the normals are re-formed only if the network has non-GPS measurements and
         * another iteration follows (ADJ:557, ADJ:582-590) */
        update_geographic(a, B, B->est);
        compute_b(a, B, B->est);
        if (a->n_tm && !last) {
            memset(B->N, 0, psize(B->n) * sizeof(double));
            update_normals(a, B);
            if (add_constraints(a, B, CON_SIM)) return ORC_ADJUST_EXCEPTION_RAISED;
        }
    }
    memcpy(B->rig, B->est, B->n * sizeof(double));
    memcpy(B->rigvar, B->N, psize(B->n) * sizeof(double));                                 /* ADJ:2536 */
    if (a->iterations == a->set.max_iterations && fabs(a->maxCorr) > a->set.iteration_threshold)
        return ORC_ADJUST_MAX_ITERATIONS_EXCEEDED;
    return ORC_ADJUST_SUCCESS;
}

/* AdjustPhasedForward (ADJ:2756-2852) */
static int phased_forward(orc_adjustment* a) {
    for (uint32_t k = 0; k < a->n_blocks; ++k) {
        blk_t* B = &a->blk[k];
        if (solve(a, B, 1, k)) return -1;                                                 /* ADJ:2812 */
        /* UpdateEstimatesForward (ADJ:3022) */
        for (uint32_t i = 0; i < B->n; ++i) B->est[i] += B->corr[i];
        if (B->last || B->isolated) {
            update_max_corr(a, B->corr, B->n);
            memcpy(B->rig, B->est, B->n * sizeof(double));
            memcpy(B->rigvar, B->N, psize(B->n) * sizeof(double));
            if (B->last) memcpy(B->corrR, B->corr, B->n * sizeof(double));
        }
        /* ShrinkForwardMatrices (ADJ:3005) */
        if (!B->isolated && !B->first) shrink_pseudo(B, 1);
        /* CarryForwardJunctions (ADJ:3065) -> CarryStnEstimatesandVariancesForward (ADJ:998) */
        if (B->isolated || B->last) continue;
        blk_t* Nx = &a->blk[k + 1];
        if (Nx->isolated) continue;
        const double tc0 = orc_now();
        if (gather_junctions(a, B, B->N, B->est, B->jsl, B->n_jsl, B->jvar, B->jestFwd)) return -1;
        memcpy(B->jvarFwd, B->jvar, psize(3 * B->n_jsl) * sizeof(double));                 /* ADJ:1051 */
        attach_junctions(Nx, B->jsl, B->n_jsl, B->jvar, B->jestFwd);
        orc_t[T_CARRY] += orc_now() - tc0;
    }
    return 0;
}

/* AdjustPhasedReverseCombine (ADJ:3461-3590) */
static int phased_reverse_combine(orc_adjustment* a) {
    for (uint32_t kk = a->n_blocks; kk-- > 0;) {
        blk_t* B = &a->blk[kk];
        /* PrepareAdjustmentReverse (ADJ:3112) */
        if (B->isolated) continue;
        if (B->last) {
            memcpy(B->N, B->NR, psize(B->n) * sizeof(double));
            memcpy(B->est, B->orig, B->n * sizeof(double));
            if (add_constraints(a, B, CON_REV)) return -1;
        }
        int combine_required = !B->last && !B->isolated && !B->first;
        /* BackupNormals (ADJ:3170) */
        if (combine_required) memcpy(B->NR, B->N, psize(B->n) * sizeof(double));
        if (solve(a, B, 1, kk)) return -1;                                                /* ADJ:3512 */
        /* UpdateEstimatesReverse (ADJ:3678) */
        for (uint32_t i = 0; i < B->n; ++i) B->est[i] += B->corr[i];
        /* CarryReverseJunctions (ADJ:3833) */
        if (!B->isolated && !B->first) {
            blk_t* Nx = &a->blk[kk - 1];
            memcpy(Nx->N, Nx->NR, psize(Nx->n) * sizeof(double));                          /* ADJ:3852 */
            memcpy(Nx->est, Nx->orig, Nx->n * sizeof(double));                             /* ADJ:3863 */
            /* CarryStnEstimatesandVariancesReverse(next = kk-1, this = kk) (ADJ:1133) */
            if (gather_junctions(a, B, B->N, B->est, Nx->jsl, Nx->n_jsl, Nx->jvar, B->jestRev)) return -1;
            attach_junctions(Nx, Nx->jsl, Nx->n_jsl, Nx->jvar, B->jestRev);
            if (add_constraints(a, Nx, CON_REV)) return -1;                                /* ADJ:3880 */
            /* PrepareAdjustmentCombine (ADJ:3336) */
            if (combine_required) {
                memcpy(B->est, B->orig, B->n * sizeof(double));                            /* ADJ:3385 */
                /* CarryStnEstimatesandVariancesCombine(kk-1, kk) (ADJ:3196) */
                memcpy(B->N, B->NR, psize(B->n) * sizeof(double));                         /* ADJ:3245 */
                attach_junctions(B, Nx->jsl, Nx->n_jsl, Nx->jvarFwd, Nx->jestFwd);
                if (add_constraints(a, B, CON_CMB)) return -1;                             /* ADJ:3392 */
                if (solve(a, B, 1, kk)) return -1;                                        /* ADJ:3556 */
                /* UpdateEstimatesCombine (ADJ:3718) */
                for (uint32_t i = 0; i < B->n; ++i) B->est[i] += B->corr[i];
                shrink_pseudo(B, 2);
            }
        }
        /* UpdateEstimatesFinal (ADJ:3744) */
        if (B->last) {
            memcpy(B->corr, B->corrR, B->n * sizeof(double));
            continue;
        }
        if (B->first) shrink_pseudo(B, 1);
        update_max_corr(a, B->corr, B->n);
        memcpy(B->rig, B->est, B->n * sizeof(double));
        memcpy(B->rigvar, B->N, psize(B->n) * sizeof(double));
        memcpy(B->orig, B->rig, B->n * sizeof(double));
    }
    return 0;
}

/* UpdateAdjustment(iterate = true) for phased mode (ADJ:473-592) */
static int phased_update_adjustment(orc_adjustment* a) {
    for (uint32_t k = 0; k < a->n_blocks; ++k) {
        blk_t* B = &a->blk[k];
        if (B->last) {
            memcpy(B->est, B->rig, B->n * sizeof(double));
            memcpy(B->orig, B->rig, B->n * sizeof(double));
        }
        update_geographic(a, B, B->est);                                                 /* ADJ:530-531 */
        compute_b(a, B, B->est);
        memset(B->N, 0, psize(B->n) * sizeof(double));
        update_normals(a, B);
        memcpy(B->NR, B->N, psize(B->n) * sizeof(double));
        if (add_constraints(a, B, CON_FWD)) return -1;
    }
    return 0;
}

/* the two passes of an iteration on their own: the CPU baseline runs them side by side on two adjustments (bench.py), the way
 * the reference's multi-thread mode overlaps its forward pass with its reverse + combination pass */
int orc_adjust_forward_pass(orc_adjustment* a) {
    if (!a->phased) return -1;
    a->maxCorr = 0.0;
    return phased_forward(a);
}
int orc_adjust_reverse_pass(orc_adjustment* a) {
    if (!a->phased) return -1;
    return phased_reverse_combine(a);
}

int orc_adjust_iteration(orc_adjustment* a) {
    a->maxCorr = 0.0;
    if (!a->phased) {
        blk_t* B = &a->blk[0];
        if (solve(a, B, 1, 0)) return -1;
        return 0;
    }
    if (phased_forward(a)) return -1;
    if (phased_reverse_combine(a)) return -1;
    return 0;
}

/* AdjustPhased (ADJ:2579-2670) */
static int adjust_phased(orc_adjustment* a) {
    a->iterations = 0;
    for (uint32_t i = 0; i < a->set.max_iterations; ++i) {
        a->maxCorr = 0.0;
        a->iterations++;
        if (phased_forward(a)) return ORC_ADJUST_EXCEPTION_RAISED;
        if (phased_reverse_combine(a)) return ORC_ADJUST_EXCEPTION_RAISED;
        if (update_iteration_diagnostics(a)) return ORC_ADJUST_EXCEPTION_RAISED;          /* ADJ:2631 */
        if (a->iterations <= 64) a->max_corr_hist[a->iterations - 1] = a->maxCorr;
        if (!(fabs(a->maxCorr) > a->set.iteration_threshold)) break;                      /* ADJ:2639 */
        if (phased_update_adjustment(a)) return ORC_ADJUST_EXCEPTION_RAISED;
    }
    if (a->iterations == a->set.max_iterations && fabs(a->maxCorr) > a->set.iteration_threshold)
        return ORC_ADJUST_MAX_ITERATIONS_EXCEEDED;                                        /* ADJ:2526-2528 */
    return ORC_ADJUST_SUCCESS;
}

int orc_adjust_run(orc_adjustment* a) { return a->phased ? adjust_phased(a) : adjust_simultaneous(a); }

/* AdjustPhasedBlock1 (ADJ:2675-2717) = one AdjustPhasedReverse (ADJ:3594-3671): every block solved "in isolation" with the junctions
 * carried from the blocks after it, so that block 1 comes out rigorous; UpdateEstimatesFinal (ADJ:3744) for every block but the
 * last (which returns at once in this mode, ADJ:3750-3753); the largest correction is block 1's only (ADJ:2705) */
int orc_adjust_run_block1(orc_adjustment* a) {
    if (!a->phased) return ORC_ADJUST_EXCEPTION_RAISED;
    a->iterations = 1;
    a->maxCorr = 0.0;
    for (uint32_t kk = a->n_blocks; kk-- > 0;) {
        blk_t* B = &a->blk[kk];
        if (B->isolated) continue;                                                        /* PrepareAdjustmentReverse: nothing to do */
        if (B->last) {
            memcpy(B->N, B->NR, psize(B->n) * sizeof(double));                             /* ADJ:3112-3168 */
            memcpy(B->est, B->orig, B->n * sizeof(double));
            if (add_constraints(a, B, CON_REV)) return ORC_ADJUST_EXCEPTION_RAISED;
        }
        if (solve(a, B, 1, kk)) return ORC_ADJUST_EXCEPTION_RAISED;                        /* ADJ:3639 */
        for (uint32_t i = 0; i < B->n; ++i) B->est[i] += B->corr[i];                       /* UpdateEstimatesReverse */
        if (!B->first) {                                                                   /* CarryReverseJunctions (ADJ:3833) */
            blk_t* Nx = &a->blk[kk - 1];
            memcpy(Nx->N, Nx->NR, psize(Nx->n) * sizeof(double));
            memcpy(Nx->est, Nx->orig, Nx->n * sizeof(double));
            if (gather_junctions(a, B, B->N, B->est, Nx->jsl, Nx->n_jsl, Nx->jvar, B->jestRev)) return ORC_ADJUST_EXCEPTION_RAISED;
            attach_junctions(Nx, Nx->jsl, Nx->n_jsl, Nx->jvar, B->jestRev);
            if (add_constraints(a, Nx, CON_REV)) return ORC_ADJUST_EXCEPTION_RAISED;
        }
        if (B->last) continue;                                                             /* UpdateEstimatesFinal returns (ADJ:3750-3753) */
        if (B->first) shrink_pseudo(B, 1);
        memcpy(B->rig, B->est, B->n * sizeof(double));
        memcpy(B->rigvar, B->N, psize(B->n) * sizeof(double));
        memcpy(B->orig, B->rig, B->n * sizeof(double));
    }
    a->maxCorr = max_value(a->blk[0].corr, a->blk[0].n);                                   /* ADJ:2705 */
    a->max_corr_hist[0] = a->maxCorr;
    return fabs(a->maxCorr) > a->set.iteration_threshold ? 2 /* ADJUST_THRESHOLD_EXCEEDED */ : ORC_ADJUST_SUCCESS;
}

uint32_t orc_adjust_iterations(const orc_adjustment* a) { return a->iterations; }
double orc_adjust_max_correction(const orc_adjustment* a, uint32_t it) {
    return (it >= 1 && it <= a->iterations && it <= 64) ? a->max_corr_hist[it - 1] : 0.0;
}
uint32_t orc_adjust_block_unknowns(const orc_adjustment* a, uint32_t b) { return a->blk[b].n; }
const uint32_t* orc_adjust_block_stations(const orc_adjustment* a, uint32_t b, uint32_t* count) {
    if (count) *count = a->blk[b].n_stn;
    return a->blk[b].stations;
}
const double* orc_adjust_block_estimates(const orc_adjustment* a, uint32_t b) { return a->blk[b].rig; }
const double* orc_adjust_block_variances(const orc_adjustment* a, uint32_t b) { return a->blk[b].rigvar; }
const double* orc_adjust_block_normals(const orc_adjustment* a, uint32_t b) { return a->blk[b].N; }
const double* orc_adjust_block_b(const orc_adjustment* a, uint32_t b, uint32_t* rows) {
    if (rows) *rows = a->blk[b].b_rows;
    return a->blk[b].b;
}
const double* orc_adjust_weights(const orc_adjustment* a) { return a->W; }
const char* orc_adjust_error(const orc_adjustment* a) { return a->err; }
void orc_adjust_solve_stats(const orc_adjustment* a, uint64_t* solves, double* sum_n3) {
    if (solves) *solves = a->solves;
    if (sum_n3) *sum_n3 = a->sum_n3;
}

/* ========================================================================== */
/* post-adjustment statistics                                                  */
/* ========================================================================== */
#define ORC_UNRELIABLE 999.99   /* include/config/dnaconsts.hpp:119 */
#define ORC_STABLE_LIMIT 700.0  /* include/config/dnaconsts.hpp:120 */

int orc_adjust_statistics(orc_adjustment* a, double critical_value, orc_statistics* out) {
    const orc_network* net = &a->net;
    const size_t ncomp = (size_t)net->n_baselines * 3;
    for (int f = 0; f < 7; ++f) {
        free(a->msr_field[f]);
        a->msr_field[f] = (double*)calloc(ncomp + 1, sizeof(double));
    }
    double *measAdj = a->msr_field[0], *measCorr = a->msr_field[1], *adjPrec = a->msr_field[2], *resPrec = a->msr_field[3],
           *nstat = a->msr_field[4], *pelzer = a->msr_field[5], *measPrec = a->msr_field[6];
    /* a-priori variances as the records hold them after scaling (SetGPSVarianceMatrix, ADJ:4282) */
    {
        size_t voff = 0;
        for (uint32_t c = 0; c < a->n_clusters; ++c) {
            uint32_t i0 = a->cl_off[c], k = a->cl_off[c + 1] - i0, nc = 3 * k;
            for (uint32_t j = 0; j < k; ++j)
                for (int e = 0; e < 3; ++e) {
                    if (net->n_clusters)
                        measPrec[3 * (size_t)(i0 + j) + e] = net->cluster_vcv[voff + (size_t)(3 * j + e) * nc + 3 * j + e];
                    else
                        measPrec[3 * (size_t)(i0 + j) + e] = net->vcv6[(size_t)(i0 + j) * 6 + sym6(e, e)];
                }
            voff += (size_t)nc * nc;
        }
    }
    for (int f = 0; f < 8; ++f) {
        free(a->tm_field[f]);
        a->tm_field[f] = (double*)calloc((size_t)a->n_tm + 1, sizeof(double));
    }
    double chi_total = 0.0;
    uint32_t outliers = 0, msr_params = 0;
    for (uint32_t blk = 0; blk < a->n_blocks; ++blk) {
        blk_t* B = &a->blk[blk];
        update_geographic(a, B, B->rig);                          /* UpdateAdjustment(false): ADJ:496-531, ADJ:541-545 */
        compute_b(a, B, B->rig);                                  /* ... and ADJ:549 */
        free(B->prec);
        B->prec = (double*)calloc((size_t)B->m * 2 + 1, sizeof(double));
        const double* V = B->rigvar;
        double chi = 0.0;                                         /* ComputeChiSquare (ADJ:7257) starts from zero per block */
        uint32_t row = 0, prow = 0, trow = 0;
        for (uint32_t c = 0; c < B->n_cml; ++c) {
            const int64_t t = tm_index(a, B->cml[c]);
            if (t >= 0) {
                /* ComputePrecisionAdjMsrs_A / _BCEKLMSVZ / _HIJPQR (ADJ:7877-8007): a S a^T */
                const double* dr = B->trow + 9 * (size_t)trow++;
                const int ns = tm_station_count(net->t_type[t]);
                uint32_t loc[3];
                for (int q = 0; q < ns; ++q) loc[q] = 3 * local_index(B, net->t_stn[3 * (size_t)t + q]);
                double part[9], prec = 0.0;
                for (int s_ = 0; s_ < ns; ++s_)
                    for (int i = 0; i < 3; ++i) {
                        double acc = 0.0;
                        for (int j = 0; j < ns; ++j)
                            for (int e = 0; e < 3; ++e) acc += dr[3 * j + e] * packed_get(V, B->n, loc[j] + e, loc[s_] + i);
                        part[3 * s_ + i] = acc;
                    }
                for (int s_ = 0; s_ < ns; ++s_)
                    for (int i = 0; i < 3; ++i) prec += part[3 * s_ + i] * dr[3 * s_ + i];
                B->prec[prow++] = prec;
                /* UpdateMsrRecord (ADJ:8187) */
                double corr = -B->b[row], adj = a->t_val[t] + corr;
                switch (net->t_type[t]) {
                    case 'E': {
                        const uint32_t g1 = net->t_stn[3 * (size_t)t], g2 = net->t_stn[3 * (size_t)t + 1];
                        const double* X1 = B->rig + loc[0];
                        const double* X2 = B->rig + loc[1];
                        double r = radius_in_chord_direction(X1, X2, a->geo[3 * (size_t)g1], a->geo[3 * (size_t)g1 + 1], a->geo[3 * (size_t)g2]);
                        adj = asin(adj / 2.0 / r) * 2.0 * r;
                        break;
                    }
                    case 'M': {
                        const uint32_t g1 = net->t_stn[3 * (size_t)t], g2 = net->t_stn[3 * (size_t)t + 1];
                        adj = ellipsoid_chord_to_msl_arc(adj, a->geo[3 * (size_t)g1], a->geo[3 * (size_t)g2], net->stn_geoid[g1], net->stn_geoid[g2]);
                        break;
                    }
                    case 'H': case 'L': case 'V': adj -= a->t_corr[t]; break;
                    case 'A': case 'I': case 'J': case 'K': case 'Z': adj += a->t_corr[t]; break;
                    case 'D':   /* UpdateMsrRecord (ADJ:8194-8199, 8255-8261) */
                        if (adj > ORC_TWO_PI) adj -= ORC_TWO_PI;
                        adj += a->t_corr[t];
                        break;
                    default: break;
                }
                const double mp = net->t_var[t];
                double rp = mp - prec;
                if (rp < 0.0) rp = fabs(rp);
                double pz = sqrt(mp) / sqrt(rp);
                if (pz < 0.0 || pz > ORC_STABLE_LIMIT) pz = ORC_UNRELIABLE;
                a->tm_field[0][t] = adj;
                a->tm_field[1][t] = corr;
                a->tm_field[2][t] = prec;
                a->tm_field[3][t] = rp;
                a->tm_field[4][t] = corr / sqrt(rp);
                a->tm_field[5][t] = pz;
                a->tm_field[6][t] = mp;
                a->tm_field[7][t] = a->t_corr[t];
                if (fabs(a->tm_field[4][t]) > critical_value) outliers++;
                chi += B->b[row] * B->b[row] / net->t_var[t];     /* ComputeChiSquare_ABCEHIJKLMPQRSVZ (ADJ:8430) */
                row += 1;
                continue;
            }
            const uint32_t cl = B->cml[c], i0 = a->cl_off[cl], k = a->cl_off[cl + 1] - i0, nc = 3 * k;
            /* ComputePrecisionAdjMsrs_GX (ADJ:8009) / _Y (ADJ:8037) */
            for (uint32_t j = 0; j < k; ++j) {
                const uint32_t i = i0 + j;
                const uint32_t s2 = 3 * local_index(B, net->stn2[i]);
                const int point = net->stn1[i] == 0xffffffffu;
                const uint32_t s1 = point ? 0 : 3 * local_index(B, net->stn1[i]);
                for (int r = 0; r < 3; ++r)
                    for (int q = r; q < 3; ++q, ++prow) {
                        if (point)
                            B->prec[prow] = packed_get(V, B->n, s2 + r, s2 + q);
                        else {
                            /* Precision_Adjusted_GNSS_bsl (dnatemplatematrixfuncs.hpp:255-297) */
                            double tmp = (0.0 - packed_get(V, B->n, s1 + r, s1 + q)) + packed_get(V, B->n, s2 + r, s1 + q);
                            double tmpk = (0.0 - packed_get(V, B->n, s1 + r, s2 + q)) + packed_get(V, B->n, s2 + r, s2 + q);
                            B->prec[prow] = tmpk - tmp;
                        }
                    }
                /* UpdateMsrRecords_GXY (ADJ:8152) -> UpdateMsrRecord (ADJ:8187), UpdateMsrRecordStats (ADJ:8291) */
                static const int diag6[3] = {0, 3, 5};
                for (int e = 0; e < 3; ++e) {
                    const size_t g = 3 * (size_t)i + e;
                    measCorr[g] = -B->b[row + 3 * j + e];
                    measAdj[g] = net->obs[g] + measCorr[g];
                    adjPrec[g] = B->prec[prow - 6 + diag6[e]];
                    resPrec[g] = measPrec[g] - adjPrec[g];
                    if (resPrec[g] < 0.0) resPrec[g] = fabs(resPrec[g]);
                    pelzer[g] = sqrt(measPrec[g]) / sqrt(resPrec[g]);
                    if (pelzer[g] < 0.0 || pelzer[g] > ORC_STABLE_LIMIT) pelzer[g] = ORC_UNRELIABLE;
                    nstat[g] = measCorr[g] / sqrt(resPrec[g]);
                    if (fabs(nstat[g]) > critical_value) outliers++;
                }
            }
            /* chi-square */
            const double* r = B->b + row;
            if (!a->cl_W[cl]) {
                /* ComputeChiSquare_G (ADJ:8530) */
                const double* w6 = a->W + (size_t)i0 * 6;
                double cs = 0.0;
                for (int rr = 0; rr < 3; ++rr)
                    for (int cc = 0; cc < 3; ++cc) cs += w6[sym6(rr, cc)] * r[rr] * r[cc];
                chi += cs;
            } else {
                /* ComputeChiSquare_XY (ADJ:8551): (r^T W) r */
                const double* W = a->cl_W[cl];
                double tot = 0.0;
                for (uint32_t col = 0; col < nc; ++col) {
                    double rv = 0.0;
                    for (uint32_t rr = 0; rr < nc; ++rr) rv += r[rr] * W[(size_t)col * nc + rr];
                    tot += rv * r[col];
                }
                chi += tot;
            }
            row += nc;
        }
        chi_total += chi;                                         /* ComputeChiSquareNetwork (ADJ:7315) */
        msr_params += B->m;
    }
    /* ComputeGlobalPelzer (ADJ:8302) / _GXY (ADJ:8396) */
    double sum = 0.0;
    uint32_t num = 0;
    for (uint32_t blk = 0; blk < a->n_blocks; ++blk) {
        blk_t* B = &a->blk[blk];
        for (uint32_t c = 0; c < B->n_cml; ++c) {
            const int64_t t = tm_index(a, B->cml[c]);
            if (t >= 0) {
                double* p = &a->tm_field[5][t];                                /* ADJ:8338: "< STABLE_LIMIT" for these types */
                if (*p > 0.0 && *p < ORC_STABLE_LIMIT) {
                    sum += (*p * *p - 1.0);
                    num++;
                } else
                    *p = ORC_UNRELIABLE;
                continue;
            }
            for (uint32_t i = a->cl_off[B->cml[c]]; i < a->cl_off[B->cml[c] + 1]; ++i)
                for (int e = 0; e < 3; ++e) {
                    double* p = &pelzer[3 * (size_t)i + e];
                    if (*p > 0.0 && *p < ORC_UNRELIABLE) {
                        sum += (*p * *p - 1.0);
                        num++;
                    } else
                        *p = ORC_UNRELIABLE;
                }
        }
    }
    /* unknown parameters: 3 per station minus the constrained components (ADJ:647-672) */
    uint32_t unknowns = 3 * net->n_stations;
    for (uint32_t s = 0; s < net->n_stations; ++s)
        for (int c = 0; c < 3; ++c)
            if (net->constraints[3 * (size_t)s + c] == 'C') unknowns--;
    out->chi_squared = chi_total;
    out->measurement_params = msr_params;
    out->unknown_params = unknowns;
    out->dof = (int)msr_params - (int)unknowns;                   /* ComputeGlobalNetStat (ADJ:6854) */
    out->sigma_zero = out->dof != 0 ? chi_total / out->dof : 0.0;
    out->global_pelzer = num ? sqrt(sum / num) : ORC_UNRELIABLE;
    out->potential_outliers = outliers;
    return 0;
}

const double* orc_adjust_msr_field(const orc_adjustment* a, int field) { return (field >= 0 && field < 7) ? a->msr_field[field] : NULL; }
const double* orc_adjust_block_prec_adj_msrs(const orc_adjustment* a, uint32_t b, uint32_t* rows) {
    if (rows) *rows = (a->blk[b].m - a->blk[b].n_trow) * 2 + a->blk[b].n_trow;
    return a->blk[b].prec;
}
const double* orc_adjust_tmsr_field(const orc_adjustment* a, int field) { return (field >= 0 && field < 8) ? a->tm_field[field] : NULL; }
const double* orc_adjust_station_llh(const orc_adjustment* a) { return a->geo; }
void orc_tmsr_evaluate(orc_adjustment* a, uint32_t t, const double* xyz9, double* computed, double* row9) {
    tm_evaluate(a, t, xyz9, xyz9 + 3, xyz9 + 6, computed, row9);
}
