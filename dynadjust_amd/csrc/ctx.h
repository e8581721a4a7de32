// Internal definition of the device context behind include/dnagpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <stdlib.h>
#include <unordered_map>
#include <vector>
#include "../../include/dnagpu.h"
#include "sym_inverse.h"

struct dnagpu_matrix {
    double* F = nullptr;     // np_max x np_max storage, used with ld = np
    double* jest = nullptr;  // junction estimates attached to the matrix (n_max doubles)
    double* jrhs = nullptr;  // information form (dnagpu_schur_carry): the reduced right-hand side that goes with F; allocated on first use
    int form = 0;            // 0: jest = the junction stations' ADJUSTED estimates (weights F); 1: jest = the estimates F and jrhs were formed at
    uint32_t n_max = 0, np_max = 0;
    uint32_t n = 0, np = 0;  // current logical order / padded order
};

// State of a block between dnagpu_block_reduce(..., keep) and dnagpu_partial_complete: see include/dnagpu.h
// (the permuted normals -> factor pieces -> inverse in the elimination's order never outlive a call: they live in the chain's X workspace)
struct dnagpu_partial {
    double* X = nullptr;     // (n_cap)^2: L^-1 -- own storage, or the storage of `store` (dnagpu_partial_create_in)
    dnagpu_matrix* store = nullptr;   // the matrix whose storage X borrows between the elimination and the completion
    double* WK = nullptr;    // k_cap x n_cap: L_KI
    int32_t* map = nullptr;  // n_cap: elimination order -> natural unknown (-1 padding, -2 the rhs row)
    uint32_t n_cap = 0, k_cap = 0;               // capacities (padded orders)
    uint32_t n = 0, nj = 0, nip = 0, njp = 0, npp = 0;
    bool valid = false;      // reduce done, completion pending
    bool completed = false;  // completion done: X (the eliminated part's inverse factor) and WK still describe the block
    bool spine = false;      // light form (dnagpu_partial_create_spine): X holds the elimination's block factor (sym_inverse.h: sym_spine_async), no WK
    bool factored = false;   // dnagpu_partial_complete_factor done: X = L^-1 of the WHOLE block (elimination order); the inverse itself is pending
};

// dnagpu_small_batch_*: the device-side table of many small blocks' per-iteration steps (small_steps.h) and what it owns
struct dnagpu_small_batch {
    uint32_t n = 0;
    void* table = nullptr;                // SmallBlockDesc[n], device
    double* result = nullptr;             // 2 n doubles, device
    double* result_host = nullptr;        // ... and their page-locked landing zone
    std::vector<uint32_t*> idx_dev;       // the junction station lists, device copies owned by the batch
};

// dnagpu_chain_plan_*: chain steps as data (small_steps.h CbStep), grouped into lock-step batches
struct dnagpu_chain_plan {
    size_t n_steps = 0;
    std::vector<uint32_t> batch_first;                // n_batches + 1
    struct Shape { uint32_t nip, njp, npp, outnp_max; };
    std::vector<Shape> shape;                         // per batch: the padded orders its members are eliminated in
    void* table = nullptr;                            // CbStep[n_steps], device
    void* blob = nullptr;                             // index lists, maps, constraint blocks, scratch vectors (one allocation)
    double* factors = nullptr;                        // the steps' kept factors (one allocation)
    std::vector<double*> X;                           // per step: its factor
    struct Out { dnagpu_matrix* m; uint32_t nj; int junction; };
    std::vector<Out> out;                             // per step: where its result goes (host-side fields are set when the step runs)
    std::vector<uint8_t> factored;                    // per batch: dnagpu_chain_plan_run has been through it
    bool keeps = true;                                // the steps' factors stay (false: beyond the budget -- every run eliminates again)
    double factor_bytes = 0.0;
    double flops = 0.0;
};

// dnagpu_block_table_*: device rows (adjust_kernels.h BlockTableRow) of a set of GNSS-only blocks
struct dnagpu_block_table {
    uint32_t n = 0, max_len = 0;
    void* rows = nullptr;
};

namespace dnagpu {

struct Block {
    uint32_t n_stn = 0, n_bl = 0;
    bool wb_own = false;        // dnagpu_block_set_terrestrial has given wb[] allocations of their own (GNSS + terrestrial vectors)
    void* arena = nullptr;      // dnagpu_block_create: everything of a fixed size in one allocation (stations, the vectors of every chain, baselines)
    // stations (3*n_stn)
    // "estimated" state exists once per chain (the reference's v_*_ / v_*R_ twins,
    // dnaadjust.hpp:1340-1348) so that the forward and the reverse/combine chain can
    // work on the same block concurrently
    double *x_orig = nullptr, *x_rig = nullptr;
    double *x_est[DNAGPU_NUM_CHAINS] = {};
    double *rhs[DNAGPU_NUM_CHAINS] = {};
    double *corr[DNAGPU_NUM_CHAINS] = {};
    // baselines, SoA
    uint32_t *s1 = nullptr, *s2 = nullptr;
    double *obs = nullptr;  // 3*n_bl
    // measurement weights: 3x3 blocks (9 doubles, column-major) of every cluster's inverse variance
    // matrix; a k-vector cluster owns k*k consecutive blocks, block (j, j') at wrow(j) + j'
    double *Wblk = nullptr;
    uint32_t n_wblk = 0;
    uint32_t *vec_wrow = nullptr, *vec_c0 = nullptr, *vec_k = nullptr;   // per vector: first block of its row, first vector and size of its cluster
    double *wb[DNAGPU_NUM_CHAINS] = {};                  // 3*n_bl: W b per vector
    double *b[DNAGPU_NUM_CHAINS] = {};  // 3*n_bl, measured - computed
    // deterministic formation structure: station-pair blocks (row >= col), each
    // with the CML-ordered list of contributing baselines
    uint32_t n_pairs = 0;
    uint32_t *pair_row = nullptr, *pair_col = nullptr, *pair_off = nullptr;  // n_pairs(+1)
    uint32_t *pair_ent = nullptr;  // per contribution: weight-block index << 1 | negative
    // per-station incidence (CML order) for the rhs: entry = baseline*2 + (1 if station is stn2)
    uint32_t *inc_off = nullptr, *inc = nullptr;
    // scratch for max-correction reduction (value, index) per chain
    double* red[DNAGPU_NUM_CHAINS] = {};
    // terrestrial measurements (one design row each; csrc/terrestrial.h).  Their 3x3 blocks w a_p^T a_q and vectors
    // a_p w b change with the estimates: one copy per chain, behind the GNSS weight blocks / W b vectors
    uint32_t n_t = 0, n_tblk = 0, n_tvec = 0;
    uint8_t* t_type = nullptr;
    uint32_t *t_stn = nullptr, *t_blk0 = nullptr, *t_vec0 = nullptr;
    double *t_val = nullptr, *t_pre = nullptr, *t_var = nullptr, *t_ih = nullptr, *t_th = nullptr;
    double *s_llh = nullptr, *s_geoid = nullptr, *s_defl = nullptr;     // station records: geodetic position, N, deflections
    double* tb[DNAGPU_NUM_CHAINS] = {};                 // n_t: measured - computed
    double* trow[DNAGPU_NUM_CHAINS] = {};               // 9 n_t: design rows
    // direction sets (type D): rows of a set share a dense weight matrix; their normal-equation blocks couple every pair of
    // station slots of the set (dnagpu_block_set_direction_sets)
    uint32_t n_dsblk = 0;                                  // blocks of all sets, stored behind the per-measurement ones
    uint32_t *ds_a = nullptr, *ds_b = nullptr, *ds_pq = nullptr, *ds_w = nullptr;   // per block: rows a, b; slots p | q << 2; weight index
    uint32_t *ds_row0 = nullptr, *ds_k = nullptr, *ds_woff = nullptr;               // per terrestrial row: first row / size / weight offset of its set (k = 0: none)
    double* ds_wts = nullptr;
    struct DsEnt { uint64_t key; uint32_t pos, blk; };
    std::vector<DsEnt> h_ds_ents;
    // dnagpu_schur_carry: unknown order with the carried junction stations last, per junction list seen (forward / reverse)
    double* corr_keep = nullptr;       // dnagpu_block_keep_corrections: a solution's corrections set aside (UpdateEstimatesFinal ADJ:3755)
    uint32_t *osc_gidx = nullptr, *osc_visit = nullptr;   // dnagpu_osc_block: the stations' indices in the network, this iteration's visit record
    std::vector<uint32_t> h_schur_idx[2];
    uint32_t h_schur_nip[2] = {0, 0}, h_schur_npp[2] = {0, 0};      // (the padded orders the cached map was laid out for)
    uint32_t* schur_idx[2] = {};
    int32_t* schur_map[2] = {};
    uint32_t* schur_spos[2] = {};      // station -> position of its first unknown in that order
    std::vector<void*> retired;        // replaced schur_map / schur_idx lists, freed with the block
    // host copies kept until the pair / incidence lists are built (dnagpu_block_set_clusters)
    std::vector<uint8_t> h_ttype;
    std::vector<uint32_t> h_tstn, h_tpos, h_cpos;
};

}  // namespace dnagpu

struct dnagpu_ctx {
    int device = 0;
    int info_carry = 1;          // dnagpu_schur_carry leaves the information form (taken from the process default at dnagpu_create)
    std::mutex err_mutex;          // err / last_info: written by whichever chain's host thread fails (dnagpu_api.hip note_error)
    std::string err;
    int last_info = 0;
    hipStream_t stream[DNAGPU_NUM_CHAINS] = {};
    hipEvent_t ev[DNAGPU_NUM_CHAINS] = {};
    dnagpu::InvWorkspace ws[DNAGPU_NUM_CHAINS];
    double* plan_scratch[DNAGPU_NUM_CHAINS] = {};      // dnagpu_chain_plan_run of a plan that keeps no factors: the members' factors of one batch
    size_t plan_scratch_cap[DNAGPU_NUM_CHAINS] = {};
    double* symv_part[DNAGPU_NUM_CHAINS] = {};
    uint32_t symv_cap[DNAGPU_NUM_CHAINS] = {};
    // small per-chain staging buffers for index lists / 3x3 weights / vectors
    uint32_t* scr_u32[DNAGPU_NUM_CHAINS] = {};
    size_t scr_u32_cap[DNAGPU_NUM_CHAINS] = {};
    double* scr_f64[DNAGPU_NUM_CHAINS] = {};
    size_t scr_f64_cap[DNAGPU_NUM_CHAINS] = {};
    // index lists that keep coming back (a block's kept / junction stations, every chain step of every iteration): their device copies,
    // per chain, found by content -- no upload and no stream synchronisation from the second use on (dnagpu_api.hip stage_u32)
    struct IndexList {
        std::vector<uint32_t> host;
        uint32_t* dev = nullptr;
    };
    std::unordered_multimap<uint64_t, IndexList> idx_cache[DNAGPU_NUM_CHAINS];
    // ... and the constraint weights that come with such lists (stage_f64: 9 doubles per station, the same in every iteration)
    struct ValueList {
        std::vector<double> host;
        double* dev = nullptr;
    };
    std::unordered_multimap<uint64_t, ValueList> val_cache[DNAGPU_NUM_CHAINS];
    // pinned host landing zone for (max correction, row)
    double* red_val_host[DNAGPU_NUM_CHAINS] = {};
    uint32_t* red_idx_host[DNAGPU_NUM_CHAINS] = {};
    int* bad_dev = nullptr;
    std::map<uint32_t, dnagpu::Block> blocks;
    // dnagpu_matrix_download_packed_async: per chain a copy stream, a device staging buffer (the packed triangle) and the events
    // that order pack -> copy -> next pack; created on first use
    hipStream_t copy_stream[DNAGPU_NUM_CHAINS] = {};
    hipEvent_t pack_done[DNAGPU_NUM_CHAINS] = {}, copy_done[DNAGPU_NUM_CHAINS] = {};
    double* stage_buf[DNAGPU_NUM_CHAINS] = {};
    size_t stage_cap[DNAGPU_NUM_CHAINS] = {};
    bool copy_pending[DNAGPU_NUM_CHAINS] = {};
    std::mutex schur_mutex;        // the per-block unknown orders of dnagpu_schur_carry are created on first use, by either chain's thread
    int dist_rank = 0, dist_world = 1;            // intra-block distributed inverse (dnagpu_set_inverse_exchange)
    dnagpu_exchange_fn exchange = nullptr;
    void* exchange_user = nullptr;
    // dnagpu_profile_hbm_*: HIP events around every launch of the large HBM-bound kernels (kinds: dnagpu.h DNAGPU_HBM_*), per chain
    struct HbmRec {
        int kind;
        double bytes;
        hipEvent_t e0, e1;
    };
    // dnagpu_osc_*: per station of the network (UpdateIterationDiagnostics' corrPrev_ / stnOscCount_, ADJ:7472-7507)
    double* osc_prev = nullptr;
    uint32_t *osc_seen = nullptr, *osc_cnt = nullptr, *osc_flagged = nullptr;
    size_t osc_stations = 0;
    // dnagpu_osc_blocks: the blocks' rows (adjust_kernels.h OscRow) and the stations' visit lists of its one launch, kept while the blocks stay the same
    void* osc_rows = nullptr;
    uint32_t* osc_off = nullptr;
    void* osc_visits = nullptr;
    uint64_t osc_key = 0;
    std::vector<uint32_t> osc_blks;          // the block ids the visit lists were built for (osc_key is their hash)
    std::vector<uint8_t> osc_rows_host;      // the rows as uploaded last
    bool hbm_profile = false;
    std::vector<HbmRec> hbm_recs[DNAGPU_NUM_CHAINS];
    std::vector<hipEvent_t> hbm_free[DNAGPU_NUM_CHAINS];
    double hbm_bytes[8] = {}, hbm_ms[8] = {};
    uint64_t hbm_count[8] = {};
    bool profile = false;
    double profile_ms_acc = 0.0;   // union length of the timed GEMM runs collected so far (dnagpu_profile_get)
};
