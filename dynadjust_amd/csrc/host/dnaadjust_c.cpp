// C-ABI wrappers declared in include/dnaadjust_c.h.
#include "../../../include/dnaadjust_c.h"

#include <cstdio>
#include <cstring>
#include <string>

#include "dna_adjust.hpp"
#include "dnaimport_lite.hpp"
#include "dnaio.hpp"
#include "statfuncs.hpp"
#include "synth.hpp"

using dynadjust::networkadjust::dna_adjust;

struct dnaadj_handle {
    dna_adjust* adj = nullptr;
    std::string err;
};

namespace {
template <class F>
int guarded(dnaadj_handle* h, F&& f) {
    if (!h || !h->adj) return DNAADJ_EINVAL;
    try {
        f();
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        h->err = e.what();
        return DNAADJ_EXCEPTION;
    } catch (...) {
        h->err = "unknown exception";
        return DNAADJ_EXCEPTION;
    }
}
}  // namespace

extern "C" {

void dnaadj_default_settings(dnaadj_settings* s) {
    if (!s) return;
    memset(s, 0, sizeof(*s));
    s->adjust_mode = 0;
    s->max_iterations = 10;
    s->iteration_threshold = 0.0005f;
    s->free_std_dev = 10.0;
    s->fixed_std_dev = 1.0e-6;
    s->confidence_interval = 95.0f;
    s->schur_carry = 1;
    s->keep_factors = 1;
    s->dist_two_level = 1;
    s->defer_variances = 2;
    s->batch_blocks = 32;
    s->reuse_factors = 1;
    s->chain_runs = -1;
}

int dnaadj_create(dnaadj_handle** out) {
    if (!out) return DNAADJ_EINVAL;
    *out = nullptr;
    try {
        dnaadj_handle* h = new dnaadj_handle();
        h->adj = new dna_adjust();
        *out = h;
        return DNAADJ_OK;
    } catch (...) {
        return DNAADJ_EXCEPTION;
    }
}

void dnaadj_destroy(dnaadj_handle* h) {
    if (!h) return;
    delete h->adj;
    delete h;
}

const char* dnaadj_last_error(const dnaadj_handle* h) { return h ? h->err.c_str() : "null handle"; }

static void to_project_settings(const dnaadj_settings* s, dynadjust::project_settings& p) {
    p.a.bst_file = s->bst_file ? s->bst_file : "";
    p.a.bms_file = s->bms_file ? s->bms_file : "";
    p.a.seg_file = s->seg_file ? s->seg_file : "";
    p.s.asl_file = s->asl_file ? s->asl_file : "";
    p.a.adjust_mode = (uint16_t)s->adjust_mode;
    p.a.multi_thread = (uint16_t)(s->multi_thread ? 1 : 0);
    p.a.max_iterations = (uint16_t)s->max_iterations;
    p.a.iteration_threshold = s->iteration_threshold;
    p.a.free_std_dev = s->free_std_dev;
    p.a.fixed_std_dev = s->fixed_std_dev;
    p.a.scale_normals_to_unity = (uint16_t)(s->scale_normals_to_unity ? 1 : 0);
    p.a.device = s->device;
    if (s->confidence_interval > 0.0f) p.a.confidence_interval = s->confidence_interval;
    p.o._adj_msr_tstat = (uint16_t)(s->output_tstat ? 1 : 0);
    p.a.reuse_inverses = (uint16_t)(s->reuse_inverses ? 1 : 0);
    p.a.schur_carry = (uint16_t)(s->schur_carry ? 1 : 0);
    p.a.keep_factors = (uint16_t)(s->keep_factors ? 1 : 0);
    p.a.stage = (uint16_t)(s->stage ? 1 : 0);
    p.a.dist_rank = s->dist_rank;
    p.a.dist_world = s->dist_world > 0 ? s->dist_world : 1;
    if (s->n_devices > 1 && s->devices) p.a.devices.assign(s->devices, s->devices + s->n_devices);
    if (s->dist_transport) p.a.dist_transport = s->dist_transport;
    p.a.dist_two_level = (uint16_t)(s->dist_two_level ? 1 : 0);
    p.a.defer_variances = (uint16_t)(s->defer_variances < 0 ? 0 : s->defer_variances > 2 ? 2 : s->defer_variances);
    p.a.batch_blocks = (uint16_t)(s->batch_blocks < 0 ? 0 : s->batch_blocks > DNAGPU_BATCH_MAX ? DNAGPU_BATCH_MAX : s->batch_blocks);
    p.a.reuse_factors = (uint16_t)(s->reuse_factors ? 1 : 0);
    p.a.chain_runs = s->chain_runs < 0 ? -1 : s->chain_runs > 4096 ? 4096 : s->chain_runs;
    if (s->network_name) p.g.network_name = s->network_name;
    if (s->output_folder) p.g.output_folder = s->output_folder;
}

int dnaadj_prepare(dnaadj_handle* h, const dnaadj_settings* s) {
    if (!s) return DNAADJ_EINVAL;
    return guarded(h, [&] {
        dynadjust::project_settings p;
        to_project_settings(s, p);
        h->adj->PrepareAdjustment(p);
    });
}

int dnaadj_plan_distributed(dnaadj_handle* h, const dnaadj_settings* s, int world, double hbm_bytes, char* json, size_t cap, size_t* needed) {
    if (!s) return DNAADJ_EINVAL;
    return guarded(h, [&] {
        dynadjust::project_settings p;
        to_project_settings(s, p);
        const std::string text = h->adj->PlanDistributed(p, world, hbm_bytes);
        if (needed) *needed = text.size() + 1;
        if (json && cap) {
            const size_t n = std::min(cap - 1, text.size());
            memcpy(json, text.data(), n);
            json[n] = 0;
        }
    });
}

int dnaadj_adjust(dnaadj_handle* h, int* status) {
    return guarded(h, [&] {
        int st = (int)h->adj->AdjustNetwork();
        if (status) *status = st;
    });
}

int dnaimport_text(const char* stn_file, const char* msr_file, const char* out_base, dnaimport_summary* out, char* err, size_t errlen) {
    return dnaimport_text_geo(stn_file, msr_file, nullptr, out_base, out, err, errlen);
}

int dnaimport_text_geo(const char* stn_file, const char* msr_file, const char* geo_file, const char* out_base, dnaimport_summary* out, char* err,
                       size_t errlen) {
    if (!stn_file || !msr_file || !out_base) return DNAADJ_EINVAL;
    try {
        dynadjust::import::import_summary s;
        dynadjust::import::import_dna_text(stn_file, msr_file, out_base, &s, geo_file ? geo_file : "");
        if (out) {
            out->stations = s.stations;
            out->records = s.records;
            out->vectors = s.vectors;
            out->clusters = s.clusters;
            out->vectors_transformed = s.vectors_transformed;
        }
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return DNAADJ_EXCEPTION;
    }
}

int dnaadj_dist_rccl_available(void) { return dynadjust::networkadjust::rccl_available() ? 1 : 0; }

int dnaadj_dist_unique_id(unsigned char* id128, char* err, size_t errlen) {
    if (!id128) return DNAADJ_EINVAL;
    try {
        dynadjust::networkadjust::rccl_unique_id(id128);
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return DNAADJ_EXCEPTION;
    }
}

int dnaadj_dist_attach_rccl(dnaadj_handle* h, int rank, int world, const unsigned char* id128, int device) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return DNAADJ_EINVAL;
    return guarded(h, [&] { h->adj->AttachCommunicator(dynadjust::networkadjust::rccl_comm_create(rank, world, id128, device)); });
}

int dnaadj_adjust_distributed(dnaadj_handle* h, int* status) {
    if (!h || !h->adj) return DNAADJ_EINVAL;
    if (!h->adj->Distributed() && h->adj->DeviceInstances() < 2) {
        h->err = "dnaadj_adjust_distributed: the adjustment was not prepared across GPUs";
        return DNAADJ_EINVAL;
    }
    return dnaadj_adjust(h, status);
}

int dnaadj_dist_info(const dnaadj_handle* h, int* rank, int* world, char* transport, size_t len) {
    if (!h || !h->adj) return DNAADJ_EINVAL;
    dna_adjust* a = h->adj;
    const bool multi = a->DeviceInstances() > 1;
    if (rank) *rank = a->DistRank();
    if (world) *world = a->DistWorld();
    if (transport && len) snprintf(transport, len, "%s", a->DistTransport());
    (void)multi;
    return DNAADJ_OK;
}

int dnaadj_block_owner(const dnaadj_handle* h, uint32_t block) {
    if (!h || !h->adj || block >= h->adj->blockCount()) return -1;
    return h->adj->BlockOwner(block);
}

int dnaadj_exchange_stats(const dnaadj_handle* h, uint64_t* bytes, double* exchange_ms, double* chain_ms) {
    if (!h || !h->adj) return DNAADJ_EINVAL;
    if (bytes) *bytes = h->adj->ExchangedBytes();
    if (exchange_ms) *exchange_ms = h->adj->ExchangeMs();
    if (chain_ms) *chain_ms = h->adj->ChainPhaseMs();
    return DNAADJ_OK;
}

void* dnaadj_device_instance_context(dnaadj_handle* h, int r) {
    if (!h || !h->adj || r < 0 || r >= h->adj->DeviceInstances()) return nullptr;
    return h->adj->DeviceInstance(r)->deviceContext();
}

int dnaadj_debug_tcp_share_unique_id(int rank, int world, unsigned char* id128, const char* addr, int port, double timeout_s, char* err, size_t errlen) {
    try {
        dynadjust::networkadjust::tcp_share_unique_id(rank, world, id128, addr, port, timeout_s);
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return DNAADJ_EXCEPTION;
    }
}

int dnaadj_debug_cancel_instance(dnaadj_handle* h, int r) {
    if (!h || !h->adj || r < 0 || r >= h->adj->DeviceInstances()) return DNAADJ_EINVAL;
    h->adj->DeviceInstance(r)->CancelThisRankOnly();
    return DNAADJ_OK;
}

int dnaadj_device_instance_stats(dnaadj_handle* h, int r, dnaadj_instance_stats* out) {
    if (!h || !h->adj || !out || r < 0 || r >= h->adj->DeviceInstances()) return DNAADJ_EINVAL;
    dna_adjust* a = h->adj->DeviceInstance(r);
    out->rank = a->DistRank();
    out->device = a->deviceOrdinal();
    out->algorithmic_flops = a->ownAlgorithmicFlops();
    out->solve_flops = a->ownSolveFlops();
    out->solves = a->ownSolveCount();
    out->eliminations = a->ownEliminationCount();
    out->completions = a->ownCompletionCount();
    out->exchanged_bytes = a->ExchangedBytes();
    out->exchange_ms = a->ExchangeMs();
    out->chain_ms = a->ChainPhaseMs();
    out->rccl_ranks = a->CommunicatorRanks();
    return DNAADJ_OK;
}

int dnaadj_reset(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->ResetAdjustment(); });
}

int dnaadj_cancel(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->CancelAdjustment(); });
}

int dnaadj_generate_statistics(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->GenerateStatistics(); });
}

int dnaadj_get_statistics(const dnaadj_handle* h, dnaadj_statistics* out) {
    if (!h || !h->adj || !out) return DNAADJ_EINVAL;
    const dna_adjust& a = *h->adj;
    out->chi_squared = a.GetChiSquared();
    out->sigma_zero = a.GetSigmaZero();
    out->global_pelzer = a.GetGlobalPelzerRel();
    out->chi_upper_limit = a.GetChiSquaredUpperLimit();
    out->chi_lower_limit = a.GetChiSquaredLowerLimit();
    out->measurement_params = a.GetMeasurementCount();
    out->unknown_params = a.GetUnknownsCount();
    out->potential_outliers = a.GetPotentialOutlierCount();
    out->test_result = a.GetTestResult();
    out->degrees_of_freedom = a.GetDegreesOfFreedom();
    return DNAADJ_OK;
}

uint64_t dnaadj_measurement_record_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->GetMeasurementRecords().size() : 0; }

int dnaadj_measurement_records(const dnaadj_handle* h, void* records, uint64_t cap_records) {
    if (!h || !h->adj || !records) return DNAADJ_EINVAL;
    const auto& r = h->adj->GetMeasurementRecords();
    if (cap_records < r.size()) return DNAADJ_EINVAL;
    if (!r.empty()) memcpy(records, r.data(), r.size() * sizeof(r[0]));
    return DNAADJ_OK;
}

uint64_t dnaadj_block_prec_adj_msrs_count(const dnaadj_handle* h, uint32_t block) {
    if (!h || !h->adj || block >= h->adj->blockCount()) return 0;
    return h->adj->GetBlockPrecAdjMsrs(block).size();
}

int dnaadj_block_prec_adj_msrs(const dnaadj_handle* h, uint32_t block, double* out, uint64_t cap) {
    if (!h || !h->adj || block >= h->adj->blockCount() || !out) return DNAADJ_EINVAL;
    const auto& p = h->adj->GetBlockPrecAdjMsrs(block);
    if (cap < p.size()) return DNAADJ_EINVAL;
    if (!p.empty()) memcpy(out, p.data(), p.size() * sizeof(double));
    return DNAADJ_OK;
}

int dnaadj_serialise_adjusted_variance_matrices(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->SerialiseAdjustedVarianceMatrices(); });
}

int dnaadj_deserialise_adjusted_variance_matrices(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->DeSerialiseAdjustedVarianceMatrices(); });
}
int dnaadj_update_binary_files(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->UpdateBinaryFiles(); });
}

double dnastat_normal_quantile(double p) { return dynadjust::stat::normal_quantile(p); }
double dnastat_chi_squared_quantile(double dof, double p) { return dynadjust::stat::chi_squared_quantile(dof, p); }

uint32_t dnaadj_block_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->blockCount() : 0; }
uint32_t dnaadj_iterations(const dnaadj_handle* h) { return h && h->adj ? h->adj->CurrentIteration() : 0; }
double dnaadj_max_correction(const dnaadj_handle* h) { return h && h->adj ? h->adj->GetMaxCorrection() : 0.0; }
double dnaadj_iteration_correction(const dnaadj_handle* h, uint32_t it) { return h && h->adj ? h->adj->GetIterationCorrection(it) : 0.0; }
uint32_t dnaadj_measurement_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->GetMeasurementCount() : 0; }
uint32_t dnaadj_unknowns_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->GetUnknownsCount() : 0; }
int dnaadj_degrees_of_freedom(const dnaadj_handle* h) { return h && h->adj ? h->adj->GetDegreesOfFreedom() : 0; }
double dnaadj_adjust_time_ms(const dnaadj_handle* h) { return h && h->adj ? h->adj->adjustTimeMs() : 0.0; }
double dnaadj_solve_flops(const dnaadj_handle* h) { return h && h->adj ? h->adj->solveFlops() : 0.0; }
uint32_t dnaadj_solve_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->solveCount() : 0; }
double dnaadj_algorithmic_flops(const dnaadj_handle* h) { return h && h->adj ? h->adj->algorithmicFlops() : 0.0; }
uint32_t dnaadj_completion_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->completionCount() : 0; }
uint32_t dnaadj_elimination_count(const dnaadj_handle* h) { return h && h->adj ? h->adj->eliminationCount() : 0; }
double dnaadj_minimal_work_flops(const dnaadj_handle* h) { return h && h->adj ? h->adj->minimalWorkFlops() : 0.0; }
uint64_t dnaadj_factor_reuses(const dnaadj_handle* h) { return h && h->adj ? h->adj->FactorReuses() : 0; }
uint64_t dnaadj_small_batch_steps(const dnaadj_handle* h) { return h && h->adj ? h->adj->SmallBatchSteps() : 0; }
int dnaadj_chain_runs(const dnaadj_handle* h) { return h && h->adj ? h->adj->ChainRuns() : 0; }
uint64_t dnaadj_chain_step_reuses(const dnaadj_handle* h) { return h && h->adj ? h->adj->ChainStepReuses() : 0; }

uint32_t dnaadj_block_station_count(const dnaadj_handle* h, uint32_t block) {
    if (!h || !h->adj || block >= h->adj->blockCount()) return 0;
    return (uint32_t)h->adj->GetBlockStationList(block).size();
}

int dnaadj_block_stations(dnaadj_handle* h, uint32_t block, uint32_t* stations) {
    return guarded(h, [&] {
        const auto& l = h->adj->GetBlockStationList(block);
        if (!l.empty()) memcpy(stations, l.data(), l.size() * sizeof(uint32_t));
    });
}

int dnaadj_block_estimates(dnaadj_handle* h, uint32_t block, double* xyz) {
    return guarded(h, [&] {
        std::vector<double> v;
        h->adj->GetBlockRigorousStations(block, v);
        if (!v.empty()) memcpy(xyz, v.data(), v.size() * sizeof(double));
    });
}

int dnaadj_block_variances_packed(dnaadj_handle* h, uint32_t block, double* packed) {
    return guarded(h, [&] {
        std::vector<double> v;
        h->adj->GetBlockRigorousVariancesPacked(block, v);
        if (!v.empty()) memcpy(packed, v.data(), v.size() * sizeof(double));
    });
}

uint32_t dnaadj_station_count(const dnaadj_handle* h) {
    if (!h || !h->adj) return 0;
    uint32_t mx = 0;
    for (uint32_t b = 0; b < h->adj->blockCount(); ++b)
        for (uint32_t s : h->adj->GetBlockStationList(b)) mx = s + 1 > mx ? s + 1 : mx;
    return mx;
}

int dnaadj_adjusted_coordinates(dnaadj_handle* h, double* xyz) {
    return guarded(h, [&] {
        std::vector<double> v;
        h->adj->GetAdjustedCoordinates(v);
        if (!v.empty()) memcpy(xyz, v.data(), v.size() * sizeof(double));
    });
}

void* dnaadj_device_context(dnaadj_handle* h) { return h && h->adj ? (void*)h->adj->deviceContext() : nullptr; }

int dnasynth_write_network(const char* dir, const char* name, const dnasynth_spec* spec, dnasynth_summary* out, char* err, size_t errlen) {
    if (!dir || !name || !spec) return DNAADJ_EINVAL;
    try {
        dynadjust::synth::Spec sp;
        sp.rows = spec->rows;
        sp.cols = spec->cols;
        sp.n_baselines = spec->n_baselines;
        sp.n_blocks = spec->n_blocks ? spec->n_blocks : 1;
        if (spec->seed) sp.seed = spec->seed;
        if (spec->initial_sigma > 0) sp.initial_sigma = spec->initial_sigma;
        sp.x_clusters = spec->x_clusters;
        sp.y_cluster = spec->y_cluster != 0;
        sp.y_llh = spec->y_llh != 0;
        sp.scalars = spec->scalars != 0;
        sp.rows_lo = spec->rows_lo;
        sp.rows_hi = spec->rows_hi;
        sp.ragged = spec->ragged;
        dynadjust::synth::Summary sm;
        dynadjust::synth::write_network(dir, name, sp, &sm);
        if (out) {
            out->stations = sm.stations;
            out->baselines = sm.baselines;
            out->measurement_rows = sm.measurement_rows;
            out->blocks = sm.blocks;
            out->max_block_unknowns = sm.max_block_unknowns;
        }
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return DNAADJ_EXCEPTION;
    }
}

int dnaio_file_summary(const char* bst, const char* bms, const char* asl, uint64_t* n_stn, uint64_t* n_msr, uint64_t* n_asl, char* err,
                       size_t errlen) {
    try {
        using namespace dynadjust;
        binary_file_meta_t meta;
        if (bst) {
            std::vector<station_t> v;
            iostreams::read_bst(bst, v, meta);
            if (n_stn) *n_stn = v.size();
        }
        if (bms) {
            std::vector<measurement_t> v;
            iostreams::read_bms(bms, v, meta);
            if (n_msr) *n_msr = v.size();
        }
        if (asl) {
            std::vector<asl_entry_t> v;
            iostreams::read_asl(asl, v);
            if (n_asl) *n_asl = v.size();
        }
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return DNAADJ_EXCEPTION;
    }
}

size_t dnaio_sizeof_station(void) { return sizeof(dynadjust::station_t); }
size_t dnaio_sizeof_measurement(void) { return sizeof(dynadjust::measurement_t); }

}  // extern "C"

// Reads a .seg file with the product's reader (SegFile::LoadSegFile restatement) and returns, per
// block, {network id, junction count, inner count, measurement count, design rows, first inner,
// first junction or UINT32_MAX, first measurement or UINT32_MAX}.
extern "C" int dnaio_seg_summary(const char* seg_path, const char* bms_path, uint32_t* n_blocks, uint32_t* per_block8, uint32_t cap_blocks,
                                 char* err, size_t errlen) {
    try {
        using namespace dynadjust;
        std::vector<measurement_t> bms;
        binary_file_meta_t meta;
        if (bms_path) iostreams::read_bms(bms_path, bms, meta);
        iostreams::seg_data_t seg;
        iostreams::read_seg(seg_path, seg, bms_path ? &bms : nullptr);
        if (n_blocks) *n_blocks = seg.blockCount;
        for (uint32_t b = 0; b < seg.blockCount && b < cap_blocks && per_block8; ++b) {
            uint32_t* o = per_block8 + 8 * (size_t)b;
            o[0] = seg.ContiguousNetList[b];
            o[1] = (uint32_t)seg.JSL[b].size();
            o[2] = (uint32_t)seg.ISL[b].size();
            o[3] = (uint32_t)seg.CML[b].size();
            o[4] = seg.measurementCount[b];
            o[5] = seg.ISL[b].empty() ? 0xffffffffu : seg.ISL[b][0];
            o[6] = seg.JSL[b].empty() ? 0xffffffffu : seg.JSL[b][0];
            o[7] = seg.CML[b].empty() ? 0xffffffffu : seg.CML[b][0];
        }
        return DNAADJ_OK;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return DNAADJ_EXCEPTION;
    }
}

// ---- per-block steps of the phased chain -------------------------------------------------
extern "C" {

int dnaadj_block_flags(const dnaadj_handle* h, uint32_t block, int* first, int* last, int* isolated) {
    if (!h || !h->adj || block >= h->adj->blockCount()) return DNAADJ_EINVAL;
    const dynadjust::blockMeta_t& m = h->adj->BlockMeta(block);
    if (first) *first = m._blockFirst;
    if (last) *last = m._blockLast;
    if (isolated) *isolated = m._blockIsolated;
    return DNAADJ_OK;
}
uint32_t dnaadj_junction_unknowns(const dnaadj_handle* h, uint32_t block) {
    return (h && h->adj && block < h->adj->blockCount()) ? h->adj->JunctionUnknowns(block) : 0;
}
size_t dnaadj_junction_payload_doubles(const dnaadj_handle* h, uint32_t block) {
    return (h && h->adj && block < h->adj->blockCount()) ? h->adj->JunctionPayloadDoubles(block) : 0;
}
int dnaadj_phased_begin_iteration(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->PhasedBeginIteration(); });
}
int dnaadj_phased_forward_block(dnaadj_handle* h, uint32_t block, double* mv) {
    return guarded(h, [&] { double v = h->adj->PhasedForwardBlock(0, block); if (mv) *mv = v; });
}
int dnaadj_phased_reverse_block(dnaadj_handle* h, uint32_t block, double* mv) {
    return guarded(h, [&] { double v = h->adj->PhasedReverseBlock(0, block); if (mv) *mv = v; });
}
int dnaadj_phased_combine_block(dnaadj_handle* h, uint32_t block, double* mv) {
    return guarded(h, [&] { double v = h->adj->PhasedCombineBlock(0, block); if (mv) *mv = v; });
}
int dnaadj_phased_finalise_block(dnaadj_handle* h, uint32_t block) {
    return guarded(h, [&] { h->adj->PhasedFinaliseBlock(0, block); });
}
int dnaadj_phased_note_correction(dnaadj_handle* h, double mv) {
    return guarded(h, [&] { h->adj->PhasedNoteCorrection(mv); });
}
int dnaadj_phased_end_iteration(dnaadj_handle* h, int* iterate) {
    return guarded(h, [&] { bool it = h->adj->PhasedEndIteration(); if (iterate) *iterate = it ? 1 : 0; });
}
int dnaadj_phased_finish(dnaadj_handle* h, int* status) {
    return guarded(h, [&] { h->adj->PhasedFinish(); if (status) *status = (int)h->adj->GetStatus(); });
}
int dnaadj_staged(const dnaadj_handle* h) { return (h && h->adj && h->adj->IsStaged()) ? 1 : 0; }
void dnaadj_dist_set_timeout(double seconds) { dynadjust::networkadjust::dist_set_collective_timeout(seconds); }
void dnaadj_debug_stall_rank(int rank, long nth_agreement, double seconds) { dynadjust::networkadjust::debug_stall_rank(rank, nth_agreement, seconds); }
size_t dnaadj_oscillation_history(const dnaadj_handle* h, double* out9, size_t cap_records) {
    if (!h || !h->adj) return 0;
    size_t i = 0;
    for (const auto& kv : h->adj->OscillationHistory()) {
        if (out9 && i < cap_records) {
            const auto& r = kv.second;
            const double v[9] = {(double)r.stnBstIdx, (double)r.firstIteration, (double)r.lastIteration, (double)r.maxCycles, r.firstMag, r.lastMag, r.lastE, r.lastN, r.lastUp};
            std::copy(v, v + 9, out9 + 9 * i);
        }
        ++i;
    }
    return i;
}
size_t dnaadj_summaries(dnaadj_handle* h, size_t limit, char* buf, size_t cap) {
    if (!h || !h->adj) return 0;
    std::ostringstream os;
    try {
        h->adj->PrintOscillationSummary(os);
        h->adj->PrintSuspectMeasurementSummary(os, limit);
    } catch (...) {
        return 0;
    }
    const std::string s = os.str();
    if (buf && cap) {
        const size_t n = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}
int dnaadj_memory_plan(const dnaadj_handle* h, double out[12]) {
    if (!h || !h->adj || !out) return -1;
    h->adj->MemoryPlan(out);
    return 0;
}
int dnaadj_condensed_schedule(const dnaadj_handle* h) { return (h && h->adj && h->adj->CondensedSchedule()) ? 1 : 0; }
double dnaadj_batched_flops(const dnaadj_handle* h) { return (h && h->adj) ? h->adj->BatchedFlops() : 0.0; }
uint64_t dnaadj_batched_block_steps(const dnaadj_handle* h) { return (h && h->adj) ? h->adj->BatchedBlockSteps() : 0; }
size_t dnaadj_condensed_payload_doubles(const dnaadj_handle* h, uint32_t block) {
    return (h && h->adj && block < h->adj->blockCount()) ? h->adj->CondensedPayloadDoubles(block) : 0;
}
int dnaadj_phased_condense_block(dnaadj_handle* h, uint32_t block) {
    return guarded(h, [&] { h->adj->CondenseBlock(0, block); });
}
int dnaadj_phased_condensed_forward(dnaadj_handle* h, uint32_t block) {
    return guarded(h, [&] { h->adj->CondensedForwardBlock(0, block); });
}
int dnaadj_phased_condensed_reverse(dnaadj_handle* h, uint32_t block) {
    return guarded(h, [&] { h->adj->CondensedReverseBlock(0, block); });
}
int dnaadj_phased_rigorous_block(dnaadj_handle* h, uint32_t block, double* mv) {
    return guarded(h, [&] { h->adj->FinishStagedCopies(); double v = h->adj->RigorousBlock(0, block); if (mv) *mv = v; });
}
int dnaadj_phased_condense_blocks(dnaadj_handle* h, const uint32_t* blocks, size_t n) {
    return guarded(h, [&] { h->adj->CondenseBlocks(std::vector<uint32_t>(blocks, blocks + n)); });
}
int dnaadj_phased_condensed_chains(dnaadj_handle* h) {
    return guarded(h, [&] { h->adj->CondensedChains(); });
}
int dnaadj_phased_rigorous_blocks(dnaadj_handle* h, const uint32_t* blocks, size_t n) {
    return guarded(h, [&] { h->adj->RigorousBlocks(std::vector<uint32_t>(blocks, blocks + n)); });
}
int dnaadj_condensed_export(dnaadj_handle* h, uint32_t block, double* buf) {
    return guarded(h, [&] { h->adj->ExportCondensed(block, buf); });
}
int dnaadj_condensed_import(dnaadj_handle* h, uint32_t block, const double* buf) {
    return guarded(h, [&] { h->adj->ImportCondensed(block, buf); });
}
int dnaadj_statistics_prepare(dnaadj_handle* h) { return guarded(h, [&] { h->adj->StatisticsPrepare(); }); }
int dnaadj_statistics_blocks(dnaadj_handle* h, const uint32_t* blocks, size_t n) {
    return guarded(h, [&] { for (size_t i = 0; i < n; ++i) h->adj->StatisticsBlock(blocks[i]); });
}
int dnaadj_statistics_get_partial(const dnaadj_handle* h, double* chi_squared, uint32_t* outliers) {
    if (!h || !h->adj) return DNAADJ_EINVAL;
    if (chi_squared) *chi_squared = h->adj->PartialChiSquared();
    if (outliers) *outliers = h->adj->PartialOutlierCount();
    return DNAADJ_OK;
}
int dnaadj_statistics_set_partial(dnaadj_handle* h, double chi_squared, uint32_t outliers) {
    return guarded(h, [&] { h->adj->SetPartials(chi_squared, outliers); });
}
int dnaadj_record_statistics_get(const dnaadj_handle* h, double* out9, uint64_t cap_records) {
    if (!h || !h->adj || !out9 || cap_records < h->adj->RecordCount()) return DNAADJ_EINVAL;
    h->adj->GetRecordStatistics(out9);
    return DNAADJ_OK;
}
int dnaadj_record_statistics_set(dnaadj_handle* h, const double* in9, uint64_t n_records) {
    if (!h || !h->adj || !in9 || n_records != h->adj->RecordCount()) return DNAADJ_EINVAL;
    return guarded(h, [&] { h->adj->SetRecordStatistics(in9); });
}
int dnaadj_statistics_finish(dnaadj_handle* h) { return guarded(h, [&] { h->adj->StatisticsFinish(); }); }
int dnaadj_junction_export(dnaadj_handle* h, int kind, uint32_t block, double* buf) {
    return guarded(h, [&] { h->adj->ExportJunction(kind, block, buf); });
}
int dnaadj_junction_import(dnaadj_handle* h, int kind, uint32_t block, const double* buf) {
    return guarded(h, [&] { h->adj->ImportJunction(kind, block, buf); });
}
int dnaadj_block_get_coords(dnaadj_handle* h, uint32_t block, int which, double* xyz) {
    return guarded(h, [&] {
        std::vector<double> v;
        h->adj->GetBlockStations(block, which, v);
        if (!v.empty()) memcpy(xyz, v.data(), v.size() * sizeof(double));
    });
}
int dnaadj_block_set_coords(dnaadj_handle* h, uint32_t block, const double* xyz) {
    return guarded(h, [&] { h->adj->SetBlockStationsAll(block, xyz); });
}
int dnaadj_block_recompute_b(dnaadj_handle* h, uint32_t block) {
    return guarded(h, [&] { h->adj->RecomputeMeasMinusComp(block); });
}

}  // extern "C"
