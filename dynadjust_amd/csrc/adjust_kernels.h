// Internal launchers for the HBM-bound adjustment kernels (see adjust_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnagpu {
void launch_weights(const double* vcv6, const uint32_t* dst, double* wblk, uint32_t n, int* bad, hipStream_t s);
void launch_cluster_blocks(const double* F, uint32_t np, uint32_t k, double* wblk, hipStream_t s);
void launch_diag_weights(const double* wblk, const uint32_t* vec_wrow, const uint32_t* vec_c0, double* w6, uint32_t n_vec, hipStream_t s);
void launch_reset_block(const double* src, double* x_orig, double* x_rig, double* const* x_est, double* const* b, int chains, bool with_b,
                        const uint32_t* s1, const uint32_t* s2, const double* obs, uint32_t n_stn, uint32_t n_bl, hipStream_t s);
// dnagpu_block_table_*: one row per block, one launch for all of them (GNSS-only blocks)
struct BlockTableRow {
    const double* init;                 // the a-priori coordinates (device), mode 0
    double* x_orig;
    double* x_rig;
    double* x_est[8];
    double* b[8];
    const uint32_t* s1;
    const uint32_t* s2;
    const double* obs;
    uint32_t n3, nb3, last, pad;
};
// mode 0 (ResetAdjustment): original = rigorous = estimated (every chain) = init.  mode 1 (UpdateAdjustment): estimated (chains 0 .. chains-1) =
// rigorous, a last block's original = rigorous.  Both: measured-minus-computed of those chains from the new estimates.
void launch_block_table(const BlockTableRow* rows, uint32_t n, uint32_t max_len, int mode, int chains, hipStream_t s);
void launch_compute_b(const uint32_t* s1, const uint32_t* s2, const double* obs, const double* xe, double* b, uint32_t n_bl, hipStream_t s);
void launch_form_normals(const uint32_t* prow, const uint32_t* pcol, const uint32_t* poff, const uint32_t* pent, const double* wblk, double* F,
                         uint32_t np, uint32_t n_pairs, uint32_t n_gnss_blk, uint32_t terr_shift, hipStream_t s);
void launch_geodetic(const double* xe, double* llh, uint32_t n_stn, hipStream_t s);
void launch_tmsr_eval(const uint8_t* type, const uint32_t* stn, const double* val, const double* pre, const double* var, const double* ih,
                      const double* th, const uint32_t* blk0, const uint32_t* vec0, const double* xe, const double* llh, const double* geoid,
                      const double* defl, double* tb, double* trow, double* tblk, double* wb, uint32_t n_bl, uint32_t n_t, hipStream_t s);
void launch_tmsr_stats(const uint8_t* type, const uint32_t* stn, const double* trow, const double* S, uint32_t nps, double* prec, uint32_t n_t,
                       hipStream_t s);
void launch_dsets(const uint32_t* ra, const uint32_t* rb, const uint32_t* pq, const uint32_t* wi, const double* wts, const double* trow,
                  double* out, uint32_t n_blocks, const uint32_t* row0, const uint32_t* kk, const uint32_t* woff, const double* tb,
                  const uint32_t* vec0, double* wb, uint32_t n_bl, uint32_t n_t, hipStream_t s);
void launch_add_diag3x3(double* F, uint32_t np, const uint32_t* stn, const double* w9, uint32_t k, double sign, hipStream_t s);
void launch_form_rhs(const double* wblk, const uint32_t* vec_wrow, const uint32_t* vec_c0, const uint32_t* vec_k, const double* b, double* wb,
                     uint32_t n_vec, const uint32_t* ioff, const uint32_t* inc, double* rhs, uint32_t n_stn, hipStream_t s);
void launch_msr_stats(const double* wblk, const uint32_t* vec_wrow, const uint32_t* vec_c0, const uint32_t* vec_k, const uint32_t* s1,
                      const uint32_t* s2, const double* b, double* wb, const double* S, uint32_t nps, double* prec6, double* chi, uint32_t n_vec,
                      hipStream_t s);
// members of a batched launch (blockIdx.z), passed by value
struct FormMember {
    double* F;
    const int32_t* map;
    int32_t* map_out;
    const uint32_t *spos, *prow, *pcol, *poff, *pent;
    const double* wblk;
    const uint32_t* con_stn;
    const double* con_w9;
    const double* rhs;
    uint32_t n_pairs, n_gnss_blk, terr_shift, n_con, rhs_row;
};
struct FormBatch { FormMember m[32]; };
static_assert(sizeof(FormBatch) <= 3968, "the members of a batch travel as kernel arguments (4 KB)");
void launch_form_ordered_batch(const FormBatch& fb, int nb, uint32_t ld, uint32_t npp, hipStream_t s);
struct ExtractMember {
    const double* T;
    double *X, *S, *r;
    uint32_t nj, npj;
};
struct ExtractBatch { ExtractMember m[32]; };
void launch_extract_batch(const ExtractBatch& eb, int nb, uint32_t nip, uint32_t npp, uint32_t npj_max, hipStream_t s);
struct RhsMember {
    const double* wblk;
    const uint32_t *vec_wrow, *vec_c0, *vec_k;
    const double* b;
    double* wb;
    const uint32_t *ioff, *inc;
    double* rhs;
    uint32_t n_vec, n_stn;
};
struct RhsBatch { RhsMember m[32]; };
void launch_form_rhs_batch(const RhsBatch& rb, int nb, hipStream_t s);
struct UnpermuteMember {
    const double* F;
    const int32_t* map;
    double* inv;
    uint32_t n, np;
};
struct UnpermuteBatch { UnpermuteMember m[32]; };
void launch_unpermute_batch(const UnpermuteBatch& ub, int nb, uint32_t npp, hipStream_t s);
struct OscRow {
    const double* corr;
    const uint32_t* gidx;
    uint32_t* visit;
    uint32_t n_stn;
};
void launch_osc_update_stations(const OscRow* rows, const uint32_t* off, const void* visits, uint32_t n_global, double* prev, uint32_t* seen, uint32_t* cnt,
                                uint32_t* flagged, hipStream_t s);
void launch_osc_update(const double* corr, const uint32_t* gidx, uint32_t n_stn, double* prev, uint32_t* seen, uint32_t* cnt, uint32_t* visit,
                       uint32_t* flagged, hipStream_t s);
void launch_update_estimates(double* xe, const double* corr, uint32_t n, double* out_val, uint32_t* out_idx, hipStream_t s);
void launch_junction_gather(const double* S, uint32_t nps, const uint32_t* idx, uint32_t k, double* J, uint32_t npj, hipStream_t s);
void launch_gather_vec3(const double* x, const uint32_t* idx, uint32_t k, double* out, hipStream_t s);
void launch_scatter_add_vec3(double* y, const uint32_t* idx, uint32_t k, const double* v, hipStream_t s);
void launch_copy_vec3_indexed(double* y, const uint32_t* dst, const double* x, const uint32_t* src, uint32_t k, hipStream_t s);
void launch_junction_scatter(double* D, uint32_t npd, const uint32_t* idx, uint32_t k, const double* J, uint32_t npj, hipStream_t s);
void launch_junction_rhs(double* rhs, const double* xe, const uint32_t* idx, uint32_t k, const double* J, uint32_t npj, const double* jest,
                         const double* jr, hipStream_t s);
void launch_schur_permute(const double* src, uint32_t lds, const int32_t* map, const double* rhs, double* dst, uint32_t ldd, uint32_t npp,
                          hipStream_t s);
void launch_form_ordered(double* F, uint32_t ld, uint32_t npp, const int32_t* map, const uint32_t* spos, const uint32_t* prow, const uint32_t* pcol,
                         const uint32_t* poff, const uint32_t* pent, const double* wblk, uint32_t n_pairs, uint32_t n_gnss_blk, uint32_t terr_shift,
                         const uint32_t* con_stn, const double* con_w9, uint32_t n_con, const double* rhs, uint32_t rhs_row, hipStream_t s);
void launch_gather_map(const double* rhs, const int32_t* map, uint32_t npp, double* out, hipStream_t s);
void launch_scatter_map(const double* v, const int32_t* map, uint32_t npp, double* out, hipStream_t s);
void launch_gemv_t_lower(const double* A, uint32_t lda, uint32_t n, const double* y, double* out, hipStream_t s);
void launch_gemv_t(const double* A, uint32_t lda, uint32_t rows, uint32_t cols, const double* y, const double* base, double sign, double* out,
                   hipStream_t s);
void launch_gemv(const double* A, uint32_t lda, uint32_t rows, uint32_t cols, const double* x, double* part, uint32_t nchunks, int lower,
                 const double* base, double sign, double* out, uint32_t n_out, hipStream_t s);
void launch_partial_set_trailing(double* T, uint32_t ldt, uint32_t njp, const double* kk, uint32_t npk, uint32_t nj, hipStream_t s);
void launch_unpermute(const double* F, uint32_t ldf, uint32_t npp, const int32_t* map, double* inv, uint32_t np, hipStream_t s);
void launch_schur_extract(const double* T, uint32_t ldt, uint32_t nj, uint32_t npj, double* S, double* S2, double* r, hipStream_t s);
void launch_schur_estimates(const double* xe, const uint32_t* idx, uint32_t k, const double* delta, double* jest, hipStream_t s);
}  // namespace dnagpu
