// dna_adjust -- the drop-in boundary (L1) of the path: same entry points, argument meaning
// and error behaviour as dynadjust::networkadjust::dna_adjust of the reference
// (dynadjust/dnaadjust/dnaadjust.hpp:209-1362 of /root/reference/dynadjust/), for GNSS
// networks.  Everything numerical runs on the device through the dnagpu C-ABI
// (include/dnagpu.h); this class only schedules blocks and keeps host-side metadata.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <deque>
#include <iostream>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/dnagpu.h"
#include "dist_comm.hpp"
#include "dnaio.hpp"
#include "dnatypes.hpp"
#include "dna_printer.hpp"

namespace dynadjust {
namespace networkadjust {

// include/exception/dnaexception.hpp: NetAdjustException carries a message + block number
class NetAdjustException : public std::runtime_error {
public:
    NetAdjustException(const std::string& what, UINT32 block) : std::runtime_error(what), block_(block) {}
    UINT32 block() const { return block_; }

private:
    UINT32 block_;
};

// include/exception/dnaexception.hpp: thrown by the reference when a matrix cannot be allocated (dnaadjustprogress.cpp:81 catches it);
// here device allocations fail as NetAdjustException with the device layer's message, so this type is only ever caught
class NetMemoryException : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// math::MatrixInversionFailure (dnamatrix_contiguous.hpp:198)
class MatrixInversionFailure : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

struct block_timing_t {
    double form_ms = 0, invert_ms = 0, other_ms = 0;
};

void debug_stall_rank(int rank, long nth_agreement, double seconds);   // test hook (dna_adjust_dist.cpp)

class dna_adjust {
    friend class DynAdjustPrinter;        // (the reference's printers are friends too, dnaadjust.hpp:214-215)
public:
    dna_adjust();
    ~dna_adjust();
    dna_adjust(const dna_adjust&) = delete;
    dna_adjust& operator=(const dna_adjust&) = delete;

    // ---- reference interface (dnaadjust.hpp:259-405) -----------------------------------
    void PrepareAdjustment(const project_settings& projectSettings);   // ADJ:258
    _ADJUST_STATUS_ AdjustNetwork();                                   // ADJ:2140
    void GenerateStatistics();                                         // ADJ:6802 (chi-square / sigma-zero of GNSS rows)
    void SerialiseAdjustedVarianceMatrices();                          // ADJ:6770 (<net>-rva.mtx)
    void UpdateBinaryFiles();                                          // ADJ:445

    // dnaadjust.hpp:262.  With a.devices the other GPUs' instances of this process hear of it at once; across processes the ranks
    // agree on it at the next phase boundary (AgreeOnPhase), so that all of them leave the iteration at the same point
    inline void CancelAdjustment() {
        cancel_.store(true);
        for (const auto& p : peers_) p->cancel_.store(true);
    }
    inline bool IsCancelled() const { return cancel_.load(); }
    // test hook: the cancellation reaches THIS instance only, the way a signal reaches one process of a multi-process adjustment
    inline void CancelThisRankOnly() { cancel_.store(true); }
    inline UINT32 CurrentIteration() const { return currentIteration_; }
    inline UINT32 CurrentBlock() const { return currentBlock_; }
    inline bool IsPreparing() const { return isPreparing_; }
    inline bool IsAdjusting() const { return isAdjusting_; }
    inline bool processingForward() const { return forward_; }
    inline bool processingCombine() const { return isCombining_; }
    inline UINT32 blockCount() const { return blockCount_; }
    inline double GetMaxCorrection() const { return maxCorr_; }
    inline _ADJUST_STATUS_ GetStatus() const { return adjustStatus_; }
    inline UINT32 GetMeasurementCount() const { return measurementParams_; }
    inline UINT32 GetUnknownsCount() const { return unknownParams_; }
    inline int GetDegreesOfFreedom() const { return degreesofFreedom_; }
    inline double GetChiSquared() const { return chiSquared_; }
    inline double GetSigmaZero() const { return sigmaZero_; }
    inline UINT32 GetPotentialOutlierCount() const { return potentialOutlierCount_; }     // dnaadjust.hpp:341
    inline double GetChiSquaredUpperLimit() const { return chiSquaredUpperLimit_; }       // dnaadjust.hpp:344
    inline double GetChiSquaredLowerLimit() const { return chiSquaredLowerLimit_; }       // dnaadjust.hpp:347
    inline double GetGlobalPelzerRel() const { return globalPelzerReliability_; }         // dnaadjust.hpp:350
    inline UINT32 GetTestResult() const { return passFail_; }                             // dnaadjust.hpp:353
    inline bool IsAdjustmentQuestionable() const { return isAdjustmentQuestionable_; }
    inline bool GetAllFixed() const { return allStationsFixed_; }
    inline bool ExceptionRaised() const { return exceptionRaised_; }
    inline std::chrono::milliseconds adjustTime() const { return std::chrono::milliseconds((long long)std::llround(adjust_ms_)); }   // dnaadjust.hpp:310
    inline double adjustTimeMs() const { return adjust_ms_; }
    // progress-thread interface of dnaadjustprogress.cpp: iterations finished since the last poll, their wall times
    // (iterationQueue_ / iterationTimes_ in the reference, dnaadjust.hpp:364-380)
    // the largest correction of an iteration as the progress thread prints it (dnaadjust.hpp:355-359)
    std::string GetMaxCorrection(const UINT32& iteration) const;
    // dnaadjust.hpp:277 / :294-296 -- see dna_printer.hpp for what stands behind the printer
    DynAdjustPrinter* GetPrinter();
    // The plan PrepareAdjustment would make on each of `world` GPUs with `hbm_bytes` of memory, as JSON, WITHOUT a device: block owners, run
    // boundaries and merge order of the two-level chains, per rank the HBM budget (blocks, chains' workspaces, variance matrices or staged store,
    // kept factors, batch members) and the bytes of every exchange of an iteration.  (A dry run for a node one does not have yet, and the
    // device-free pin of the schedule in host/dna_adjust_dist.cpp.)
    std::string PlanDistributed(const project_settings& projectSettings, int world, double hbm_bytes);
    void PrintOscillationSummary(std::ostream& os = std::cout);
    // the stations UpdateIterationDiagnostics has recorded so far (dnaadjust.hpp:1277-1288 OscillationRecord), keyed by .bst index
    struct OscillationRecord {
        UINT32 stnBstIdx, firstIteration, lastIteration, maxCycles;
        double firstMag, lastMag, lastE, lastN, lastUp;
    };
    const std::map<UINT32, OscillationRecord>& OscillationHistory() const { return oscHistory_; }
    void PrintSuspectMeasurementSummary(std::ostream& os = std::cout, size_t limit = 20) const;
    bool NewMessagesAvailable();
    bool GetMessageIteration(UINT32& iteration);
    std::string GetIterationTime(const UINT32& iteration) const;
    inline int64_t LastBlockElapsedMs() const { return lastBlockElapsedMs_; }
    // dnaadjust.hpp:333-334 (--max-blas-threads): the BLAS of the reference runs on host cores; here the dense work runs on the device, so
    // the value is kept for the caller and otherwise unused
    static void SetMaxBlasThreads(int n) { max_blas_threads_ = n; }
    static int GetMaxBlasThreads() { return max_blas_threads_; }
    inline void SetExceptionRaised() { exceptionRaised_ = true; }                  // dnaadjust.hpp:271
    void LoadSegmentationFileParameters(const std::string& seg_filename);           // ADJ:10628: block count of a .seg file
    void DeSerialiseAdjustedVarianceMatrices();                                     // ADJ:6720: -rva.mtx / -pam.mtx back into the blocks
    void NoteIterationDone(double t0_ms);
    void CloseOutputFiles();                                                        // dnaadjust.hpp:297: the printer's report streams
    inline UINT32 CurrentBlockStationCount() const {
        return currentBlock_ < v_parameterStationList_.size() ? (UINT32)v_parameterStationList_[currentBlock_].size() : 0;
    }
    inline double GetIterationCorrection(UINT32 iteration) const {
        return iteration >= 1 && iteration <= iterationCorrections_.size() ? iterationCorrections_[iteration - 1] : 0.0;
    }

    // ---- results (what the reference's printers read through friend access) -------------
    const std::vector<UINT32>& GetBlockStationList(UINT32 block) const { return v_parameterStationList_.at(block); }
    void GetBlockRigorousStations(UINT32 block, std::vector<double>& xyz);
    void GetBlockRigorousVariancesPacked(UINT32 block, std::vector<double>& packed);
    // network-wide rigorous coordinates (3 per bst station; stations never adjusted keep their input value)
    void GetAdjustedCoordinates(std::vector<double>& xyz);
    // measurement records as GenerateStatistics() left them (measAdj, measCorr, measAdjPrec, residualPrec, NStat, TStat, PelzerRel)
    const std::vector<measurement_t>& GetMeasurementRecords() const { return bmsBinaryRecords_; }
    // v_precAdjMsrsFull_ of a block: 6 values (xx xy xz yy yz zz) per GNSS vector in CML order
    const std::vector<double>& GetBlockPrecAdjMsrs(UINT32 block) const {
        if (!peers_.empty() && !OwnsBlock(block)) return peers_.at(BlockOwner(block) - 1)->blocks_.at(block).prec_adj_msrs;
        return blocks_.at(block).prec_adj_msrs;
    }

    // ---- multi-GPU (not in the reference: its parallel driver is AdjustPhasedMultiThread, dnaadjust-multi.cpp:92-244) ----------
    // One process per GPU: give every process its rank (a.dist_rank / a.dist_world) and either attach a communicator before
    // PrepareAdjustment() or let PrepareAdjustment() make the RCCL one itself (the unique id travels over TCP from rank 0,
    // MASTER_ADDR / MASTER_PORT).  One process for several GPUs: list them in a.devices.  Either way AdjustNetwork(),
    // GenerateStatistics(), SerialiseAdjustedVarianceMatrices() and UpdateBinaryFiles() are then collective: call them on every
    // rank; the result files are written by rank 0.
    void AttachCommunicator(std::shared_ptr<DistComm> comm) { comm_ = comm; }
    int DistRank() const { return comm_ ? comm_->rank() : 0; }
    int DistWorld() const { return comm_ ? comm_->world() : 1; }
    const char* DistTransport() const { return comm_ ? comm_->transport() : "none"; }
    int CommunicatorRanks() const { return comm_ ? comm_->communicator_ranks() : 0; }
    int deviceOrdinal() const { return projectSettings_.a.device; }
    bool Distributed() const { return comm_ && (comm_->world() > 1 || force_distributed_); }
    // rank whose GPU holds block k's rigorous variances (and does its large steps); identical on every rank
    int BlockOwner(UINT32 k) const { return owner_.empty() ? 0 : owner_.at(k); }
    bool OwnsBlock(UINT32 k) const { return BlockOwner(k) == DistRank(); }
    uint64_t ExchangedBytes() const { return comm_ ? comm_->bytes_moved() : 0; }
    double ExchangeMs() const { return exchange_ms_; }      // host time spent in the exchange steps since ResetAdjustment()
    double ChainPhaseMs() const { return chain_ms_; }       // ... in the chains on the condensed blocks

    // ---- measurement helpers (not in the reference) ---------------------------------------
    // put every block back to its state right after PrepareAdjustment (initial coordinates, fresh
    // meas-minus-computed) so that AdjustNetwork can be timed repeatedly on resident data
    void ResetAdjustment();
    // (with a.devices: summed over the GPUs of the process)
    double solveFlops() const { return SumOverPeers([](const dna_adjust& a) { return a.solve_flops_; }); }   // sum of n^3 over Solve() calls (reference-equivalent)
    UINT32 solveCount() const { return (UINT32)SumOverPeers([](const dna_adjust& a) { return (double)a.solve_count_; }); }
    UINT32 eliminationCount() const { return (UINT32)SumOverPeers([](const dna_adjust& a) { return (double)a.elimination_count_; }); }
    UINT32 condenseCount() const { return (UINT32)SumOverPeers([](const dna_adjust& a) { return (double)a.condense_count_; }); }
    UINT32 completionCount() const { return (UINT32)SumOverPeers([](const dna_adjust& a) { return (double)a.completion_count_; }); }
    double algorithmicFlops() const { return SumOverPeers([](const dna_adjust& a) { return a.algorithmic_flops_; }); }
    // this instance's own share of the above (one rank of a multi-GPU adjustment)
    double ownAlgorithmicFlops() const { return algorithmic_flops_; }
    double minimalWorkFlops() const { return SumOverPeers([](const dna_adjust& a) { return a.min_work_flops_; }); }
    double ownSolveFlops() const { return solve_flops_; }
    UINT32 ownSolveCount() const { return solve_count_; }
    UINT32 ownEliminationCount() const { return elimination_count_; }
    UINT32 ownCompletionCount() const { return completion_count_; }
    // the instance that drives GPU r of a.devices (0 = this one)
    dna_adjust* DeviceInstance(int r) { return r == 0 ? this : peers_.at(r - 1).get(); }
    int DeviceInstances() const { return 1 + (int)peers_.size(); }
    dnagpu_ctx* deviceContext() const { return ctx_; }

private:
    struct constraint_list {
        std::vector<UINT32> stn;   // block-local station index
        std::vector<double> w9;    // 3x3 column-major each
    };
    struct block_t {
        std::vector<UINT32> stn1, stn2;       // block-local station of each baseline
        std::vector<double> obs;
        std::vector<UINT32> cluster_off;      // vectors of measurement c: cluster_off[c] .. cluster_off[c+1]-1
        std::vector<double> vcv;              // full 3k x 3k variance matrix per measurement, concatenated
        constraint_list con_fwd, con_rev, con_cmb, con_sim;
        std::vector<UINT32> jsl_here;         // local index of JSL(block) stations in this block
        std::vector<UINT32> jsl_in_next;      // local index of JSL(block) stations in block+1
        std::vector<UINT32> jslprev_here;     // local index of JSL(block-1) stations in this block
        dnagpu_matrix* jfwd = nullptr;        // v_junctionVariancesFwd_ + v_junctionEstimatesFwd_
        dnagpu_matrix* jrev = nullptr;        // v_junctionVariances_ (reverse) + v_junctionEstimatesRev_
        dnagpu_matrix* rigvar = nullptr;      // v_rigorousVariances_
        double* rig_host = nullptr;           // staged: v_rigorousVariances_ packed (lower, column-major) in page-locked host memory ...
        bool rig_on_device = false;           // ... or, where the host's memory limit ends (DecideStaging), the same packed image in HBM
        bool rig_slot_factor = false;         // the slot has room for the block's packed light factor as well (n + 256 rows): PacksItsFactor
        bool has_rigvar = false;
        // a.reuse_inverses: the inverse of the forward / reverse normals of this block (the combined one is rigvar)
        dnagpu_matrix* finv = nullptr;
        dnagpu_matrix* rinv = nullptr;
        bool has_finv = false, has_rinv = false, has_cinv = false;
        // a.schur_carry, condensed schedule: the block reduced to the stations it shares with its neighbours
        std::vector<UINT32> keep;             // block-local indices of jslprev_here + jsl_here stations, ascending
        std::vector<UINT32> c_prev, c_next;   // jslprev_here / jsl_here as positions in keep (same order as those lists)
        constraint_list con_inner;            // constraints of the eliminated stations (first appearance in both directions)
        constraint_list ccon_fwd, ccon_rev, ccon_cmb;   // con_fwd / con_rev / con_cmb of the kept stations, positions in keep
        int corr_chain = 0;                   // the chain whose corrections vector holds the block's last solution (UpdateIterationDiagnostics)
        dnagpu_partial* part = nullptr;       // a.keep_factors: the condensing step's factor, completed by the rigorous solve
        UINT32 shape_ni = 0, shape_nk = 0;    // the padded orders it is eliminated in: its own, or its bucket's (AssignBatchShapes)
        bool part_allowed = false, part_valid = false;
        // a block the HBM budget denies a kept factor: its factor is made again where it is needed, in the chain's own storage (tmpfac_)
        bool part_transient = false;
        dnagpu_partial* tpart[DNAGPU_NUM_CHAINS] = {};
        // ... unless the slot of its packed variance matrix in the staged store -- in HBM (rig_on_device) or, round 6, in page-locked host
        // memory -- has room for it (rig_slot_factor): that slot holds the factor's packed lower triangle from the condensing step to the
        // variance matrix that replaces it (fac_src: the chain's descriptor the factor was made with)
        bool fac_packed = false;
        dnagpu_partial* fac_src = nullptr;
        bool var_deferred = false;            // a.defer_variances: this iteration's inverse exists as the completed factor in `part` only
        bool part_spine = false;              // a.defer_variances = 2: the kept factor in its light form (dnagpu_partial_create_spine)
        bool part_in_rigvar = false;          // the factor's inverse waits in rigvar's storage (dnagpu_partial_create_in): rigvar has n + 256
        bool rig_direct = false;              // this iteration's rigorous solve works in rigvar itself (no copy afterwards)
        bool prefactored = false;             // a batched rigorous solve has completed the factor already (RigorousBatch)
        bool inverse_pending = false;
        bool inverse_kept = false;            // CondensedReuse(): rigvar holds this adjustment's inverse, part its factor
        // a.reuse_factors (GNSS-only networks): the block's light factor, completed in an earlier iteration of this adjustment, is intact in
        // `part` -- later iterations reduce and solve right-hand sides with it (CondenseBlock / CompleteFromPartial) and factor nothing
        bool factor_live = false;
        bool factor_reused = false;           // this iteration's condensing step took the kept factor (no flops to count in the rigorous solve)
        // a block WITHOUT a kept factor (HBM budget): the rigorous solve of iteration i, which has just made (or unpacked) the factor, also
        // reduces the right-hand side of iteration i + 1 with it -- the block's next meas-minus-computed depends on its own rigorous
        // estimates only --, so that iteration i + 1 needs no condensing step of its own: red_iter = the iteration red's vector is for
        UINT32 red_iter = 0;
        // ... and the factors of the two chain steps on the block's condensed system (0: forward, 1: reverse), kept while the budget
        // lasts (chain_fac_budget_); live: made in an earlier iteration of this adjustment
        dnagpu_partial* cfac[2] = {nullptr, nullptr};
        bool cfac_live[2] = {false, false};
        bool cfac_denied[2] = {false, false};
        dnagpu_matrix* red = nullptr;         // Schur complement onto keep + reduced right-hand side (dnagpu_block_reduce)
        std::vector<double> prec_adj_msrs;    // v_precAdjMsrsFull_ (6 per GNSS vector, then 1 per terrestrial measurement)
        // terrestrial measurements of the block (CML order among themselves)
        std::vector<char> t_type;
        std::vector<UINT32> t_stn;            // 3 per measurement, block-local
        std::vector<UINT32> t_rec;            // record index in bmsBinaryRecords_
        std::vector<double> t_val, t_pre, t_var, t_ih, t_th;
        std::vector<UINT32> t_pos, c_pos;     // CML position of every terrestrial measurement / GNSS cluster
        // direction sets: rows dset_first[s] .. + dset_size[s] - 1 of the terrestrial lists are the angles of set s (type 'D')
        std::vector<UINT32> dset_first, dset_size;
        std::vector<double> dset_w;           // their dense weight matrices (k x k, column-major), one after the other
    };

    void LoadNetworkFiles();
    void LoadSegmentationMetrics();
    void BuildSimultaneousLists();
    void CreateStnAppearanceList();
    void PrepareBlocks();
    void ParseGnssMeasurement(UINT32 block, UINT32 m, block_t& B);
    void ParseTerrestrialMeasurement(UINT32 block, UINT32 m, block_t& B, const std::vector<double>& xyz);
    UINT32 ParseDirectionSet(UINT32 block, UINT32 m, block_t& B, const std::vector<double>& xyz);   // returns the number of angles
    void FormConstraintStationVarianceMatrix(UINT32 stn, double w9[9]) const;   // ADJ:2041
    UINT32 LocalIndex(UINT32 block, UINT32 stn) const;

    void AdjustSimultaneous();           // ADJ:2413
    void AdjustPhased();                 // ADJ:2579
    void AdjustPhasedBlock1();           // ADJ:2675
    void AdjustPhasedForward();          // ADJ:2756
    void AdjustPhasedReverseCombine();   // ADJ:3461
    void AdjustPhasedMultiThreadIteration();   // dnaadjust-multi.cpp:92-244 (forward || reverse chains)
    void AdjustPhasedCondensedIteration();     // a.schur_carry: condense every block, chains on the condensed blocks, rigorous solves
    // ---- multi-GPU (dna_adjust_dist.cpp) ----
    void AdjustPhasedDistributed();            // AdjustPhased across the ranks of comm_
    static int ExchangeTrampoline(void* self, void* stream, int nparts, double* const* bufs, const size_t* counts);
    void DistributedCondensedIteration();
    void DistributedReferenceIteration();
    void ExchangeCondensed();                  // broadcast of every condensed block from its owner
    void SyncCoordinates();                    // rigorous coordinates of every block + the largest correction, on every rank
    void GenerateStatisticsDistributed();
    void CollectBlockResults(UINT32 block, std::vector<double>& packed, std::vector<double>& prec);
    void ComputeBlockOwners(bool condensed);
    void AgreeOnPhase(const char* phase, const std::function<void()>& body);   // body() everywhere, then: did any rank fail?
    double* ExchangeBuffer(size_t doubles);    // device scratch of the exchange steps
    void PrepareMultiDevice(const project_settings& p);
    void OnEveryDevice(const std::function<void(dna_adjust&)>& body);
    template <class F>
    double SumOverPeers(F f) const {
        double s = f(*this);
        for (const auto& p : peers_) s += f(*p);
        return s;
    }
    // ---- two-level condensed chains (a.dist_two_level; dna_adjust_dist.cpp) ----
    struct seg_step_t {                     // one merge of the run's running system with the next condensed block
        UINT32 dev_block = 0, n_stn = 0;    // station-less device block of the assembled system
        std::vector<UINT32> pos_prev, pos_blk, keep;
        constraint_list con;                // constraints of stations that leave the system inside the run
    };
    struct segment_t {                      // the run of blocks of one rank, condensed to the stations of its two ends
        UINT32 a = 0, b = 0;
        std::vector<UINT32> stations;       // global station ids, ascending
        std::vector<UINT32> posL, posR;     // the junction stations towards the previous / next run (junction list order)
        std::vector<UINT32> dstA, srcA, dstB, srcB;   // where the stations' coordinates come from: block a / block b
        constraint_list con_fwd, con_rev;   // direction dependent constraints of the end stations
        dnagpu_matrix* S = nullptr;
        UINT32 dev_block = 0;
        std::vector<seg_step_t> steps;      // (own run only)
        dnagpu_matrix* M[2] = {nullptr, nullptr};
    };
    std::vector<segment_t> segs_;
    bool two_level_ok_ = false;
    void ReduceRun(int c, int run);
    void PrepareTwoLevel();
    void FreeTwoLevel();
    void ReduceOwnRun();                       // level 1
    void ExchangeRuns();
    void ScanRuns();                           // level 2
    void OwnRunChains();                       // level 3
    // ---- the same three levels on ONE GPU, the runs advancing together (a.chain_runs; dna_adjust_phased.cpp LockstepChains) ----
    // Every chain step is data on the device (dnagpu_chain_plan); a level's steps of all runs go out as batches of merged launches.
    struct lock_group_t { UINT32 lo = 0, hi = 0; };          // batches lo .. hi - 1 of the plan: one step of every run of a level
    struct lock_lane_t {                                     // a sequence of such groups, each depending on the one before
        std::vector<lock_group_t> groups;
        std::vector<double> flops;                           // per group: what its eliminations cost
        std::vector<double> ref_flops;                       // ... and the reference's Solve()s its block steps stand for (n^3 each)
        std::vector<UINT32> block_steps;                     // per group: how many of its steps are chain steps on a condensed block
    };
    struct lock_stage_t { std::vector<lock_lane_t> lanes; }; // lanes of a stage are independent of each other; stages follow each other
    std::vector<lock_stage_t> lock_stages_;
    std::vector<UINT32> lock_batch_slot_;                    // per batch of the plan: run / DNAGPU_CHAIN_BATCH_MAX of its members
    dnagpu_chain_plan* lock_plan_ = nullptr;
    std::vector<dnagpu_matrix*> lock_mats_;                  // the merged systems of the runs (owned)
    bool lockstep_ok_ = false;
    bool lock_factored_ = false;                             // the plan's factors are those of this adjustment's normals
    bool lock_keeps_ = false;                                // the plan keeps its steps' factors (they fit chain_fac_budget_)
    int lock_runs_ = 0;
    // ... and the kept blocks of the rigorous solves of a many-block network as data too (matrix_only steps: RigorousBatch)
    dnagpu_chain_plan* rig_plan_ = nullptr;
    std::map<std::vector<UINT32>, size_t> rig_batches_;      // a batch's members -> its batch of rig_plan_
    bool rig_plan_denied_ = false;
    void EnsureRigorousPlan(const std::vector<std::vector<UINT32>>& groups);
    void PrepareLockstepChains();
    void FreeLockstepChains();
    bool LockstepChains();                                   // false: not run (the chains go step by step)
    std::shared_ptr<DistComm> comm_;
    bool force_distributed_ = false;           // DNAGPU_FORCE_DISTRIBUTED=1: the exchange steps also run with a single rank
    bool in_collective_ = false;               // this instance is being driven as one rank by OnEveryDevice
    bool shares_device_ = false;               // another instance of this process drives the same GPU
    bool is_peer_ = false;                     // one of the per-GPU instances of a multi-device adjustment
    std::vector<int> owner_;
    std::vector<std::unique_ptr<dna_adjust>> peers_;
    double* xbuf_dev_ = nullptr;
    double* agree_dev_ = nullptr;              // two words of device memory of AgreeOnPhase (its own: it runs between a collective and the read-back of ITS buffer)
    size_t xbuf_cap_ = 0;
    double exchange_ms_ = 0.0, chain_ms_ = 0.0;
    void PrepareCondensedBlocks();
    // what the plan of PrepareCondensedBlocks / DecideStaging has set aside is allocated here, not inside the first iteration: the chains'
    // workspaces, the per-chain factor storage of blocks without a kept factor, the staged store (page-locked host memory / packed in HBM)
    void ReserveBuffers();
    void AllocateStagedSlot(UINT32 block);
    size_t StagedSlotBytes(UINT32 block, bool* holds_factor) const;
    void AllocateChainData();
    void DecideStaging();
public:
    // ---- per-block steps of the phased chain; the drivers above and the multi-GPU orchestrator
    //      (dynadjust_amd/parallel.py through dnaadjust_c.h) are built from these ----------------
    // forward solve of block k (+ carry of its junctions to k+1); returns the signed largest correction
    double PhasedForwardBlock(int chain, UINT32 k);
    // reverse solve of block k (+ carry of JSL(k-1) to k-1); for the first block the result is rigorous
    double PhasedReverseBlock(int chain, UINT32 k);
    // combination solve of an intermediate block (needs jrev[k] and jfwd[k-1] on this device)
    double PhasedCombineBlock(int chain, UINT32 k);
    // UpdateEstimatesFinal (ADJ:3744): rigorous = estimated, rigorous variances = current inverse, original = rigorous
    void PhasedFinaliseBlock(int chain, UINT32 k);
    // ---- condensed schedule (a.schur_carry): the steps of one iteration ---------------------------------
    bool CondensedSchedule() const { return SchurCarry() && condensed_ok_; }
    double BatchedFlops() const { return batched_flops_; }      // algorithmic flops of the block steps that went through batched calls
    uint64_t BatchedBlockSteps() const { return batched_members_.load(); }      // block steps that went through batched calls (a.batch_blocks)
    // (A) independent per block: eliminate every station the block shares with no other block
    void CondenseBlock(int chain, UINT32 k);
    // (B) the forward and the reverse chain on the condensed blocks: jfwd[k] / jrev[k-1] exactly as the block-level chain leaves them
    void CondensedForwardBlock(int chain, UINT32 k);
    void CondensedReverseBlock(int chain, UINT32 k);
    // (C) independent per block: the solve whose result is rigorous (forward for a last / isolated block, reverse for a
    // first block, combination otherwise) + UpdateEstimatesFinal; returns the signed largest correction
    double RigorousBlock(int chain, UINT32 k);
    // the same for lists of blocks, spread over both chains with a.multi_thread; (B) as a whole
    void CondenseBlocks(const std::vector<UINT32>& blocks);
    void CondensedChains();
    void RigorousBlocks(const std::vector<UINT32>& blocks);
    // condensed block k as one buffer: np*np matrix + np vector, np = pad128(3 * kept stations)
    size_t CondensedPayloadDoubles(UINT32 k) const;
    void FinishStagedCopies();   // staged mode: waits for the rigorous variance matrices on their way to host memory
    void ExportCondensed(UINT32 k, double* dst);
    void ImportCondensed(UINT32 k, const double* src);
    // ---- GenerateStatistics in parts (one process per GPU: every process does the blocks whose rigorous variances it holds) ----
    void StatisticsPrepare() { UpdateAdjustment(false); StatisticsBegin(); }
    void StatisticsBegin();
    void StatisticsBlock(UINT32 block);
    void StatisticsFinish();
    double PartialChiSquared() const { return chiSquared_; }
    UINT32 PartialOutlierCount() const { return potentialOutlierCount_; }
    void SetPartials(double chi_squared, UINT32 outliers) { chiSquared_ = chi_squared; potentialOutlierCount_ = outliers; }
    size_t RecordCount() const { return bmsBinaryRecords_.size(); }
    void GetRecordStatistics(double* out9) const;
    void SetRecordStatistics(const double* in9);
    // start of an iteration on this process: maxCorr = 0 (+ iteration counter)
    void PhasedBeginIteration();
    void PhasedNoteCorrection(double mv);   // maxCorr_ update rule of ADJ:3036 / ADJ:3786
    // end of an iteration: convergence test + UpdateAdjustment; returns true when another iteration is needed
    bool PhasedEndIteration();
    void PhasedFinish();                    // ValidateandFinaliseAdjustment
    const blockMeta_t& BlockMeta(UINT32 k) const { return v_blockMeta_.at(k); }
    UINT32 JunctionUnknowns(UINT32 k) const { return (UINT32)v_JSL_.at(k).size() * 3; }
    // junction payload = np*np matrix (ld np) followed by np estimates, np = pad128(3|JSL(k)|); kind 0 = forward, 1 = reverse
    size_t JunctionPayloadDoubles(UINT32 k) const;
    void ExportJunction(int kind, UINT32 k, double* dst);         // dst: host or device memory
    void ImportJunction(int kind, UINT32 k, const double* src);
    void GetBlockStations(UINT32 k, int which, std::vector<double>& xyz);
    void SetBlockStationsAll(UINT32 k, const double* xyz);        // original = estimated (all chains) = rigorous
    void RecomputeMeasMinusComp(UINT32 k);
private:
    void ComputeStatistics();            // ADJ:7116
    void UpdateGeographicCoords();       // ADJ:8711 / ADJ:8734
    void UpdateMsrRecord(measurement_t& rec, double measCorr, double measAdjPrec, double measPrec);   // ADJ:8187
    void ForEachMeasurementComponent(const std::function<void(measurement_t&)>& fn);
    void UpdateAdjustment(bool iterate); // ADJ:473
    void ValidateandFinaliseAdjustment();// ADJ:2513

    // one Solve() (ADJ:6586): normals already formed in `m`; rhs already formed
    void SolveTry(int chain, UINT32 block, dnagpu_matrix* m);
    void AddConstraints(int chain, dnagpu_matrix* m, const constraint_list& c, int sign, UINT32 block);
    void StoreRigorousVariances(int chain, UINT32 block, dnagpu_matrix* W);
    // the matrix a block step forms its normals in: the chain's work matrix, or with a.reuse_inverses the block's own
    // resident matrix for that step (kind 0 forward, 1 reverse, 2 combination / rigorous)
    dnagpu_matrix* StepMatrix(int chain, UINT32 block, int kind);
    // a.reuse_inverses asks for the inverses of iteration 1 to be reused (GNSS-only networks: nothing but the right-hand sides
    // changes).  With the condensed schedule and kept factors that is CondensedReuse(): later iterations only reduce right-hand
    // sides, run the chains on the condensed blocks and multiply by the resident rigorous variances.  Otherwise ReuseInverses():
    // the reference's schedule with one resident inverse per block step.
    // device chains (stream + workspaces) in use: one, or with a.multi_thread DNAGPU_NUM_CHAINS (the independent block steps of
    // the condensed schedule are served by all of them; cfg3: 4.07 / 3.96 / 3.88 s per step with 2 / 3 / 4; DNAGPU_CHAINS overrides)
    int NumChains() const { return (projectSettings_.a.adjust_mode != SimultaneousMode && projectSettings_.a.multi_thread) ? mt_chains_ : 1; }
    int mt_chains_ = DNAGPU_DEFAULT_CHAINS;
    // order the rigorous variance matrix of block k is created with (spare rows when it lends its storage to the kept factor)
    // (a matrix that lends its storage to the block's kept factor holds the factor's padded shape: own order + 256, or the bucket's)
    UINT32 RigvarCapacity(UINT32 k) const {
        const UINT32 n = (UINT32)v_parameterStationList_[k].size() * 3;
        return blocks_[k].part_in_rigvar ? std::max(n + 256u, blocks_[k].shape_ni + blocks_[k].shape_nk) : n;
    }
    void AssignBatchShapes();
    bool ReuseRequested() const { return projectSettings_.a.reuse_inverses != 0 && !containsNonGPS_ && !staged_; }
    bool CondensedWanted() const {
        return projectSettings_.a.schur_carry != 0 && !projectSettings_.a.scale_normals_to_unity && projectSettings_.a.adjust_mode == PhasedMode;
    }
    bool ReuseInverses() const { return ReuseRequested() && !(CondensedWanted() && projectSettings_.a.keep_factors != 0); }
    bool SchurCarry() const { return CondensedWanted() && !ReuseInverses(); }
    bool CondensedReuse() const { return ReuseRequested() && CondensedSchedule() && projectSettings_.a.keep_factors != 0; }
    // kind: 0 forward (last block), 1 reverse (first block), 2 combination
    // returns true when the block's corrections are already there (a.defer_variances: taken from the completed factor, W untouched)
    bool CompleteFromPartial(int chain, UINT32 block, int kind, dnagpu_matrix* W);
    // a.defer_variances: the inverses that the iterations left as completed factors (block_t::var_deferred), once the iterations have ended
    bool DeferVariances() const { return projectSettings_.a.defer_variances != 0 && !ReuseRequested(); }
    // a.reuse_factors: iterations >= 2 of a GNSS-only network keep the factors of iteration 1 (the reference's own licence in
    // simultaneous mode, ADJ:2452-2457); needs the condensed schedule with light kept factors
    bool FactorReuse() const {
        return projectSettings_.a.reuse_factors != 0 && !containsNonGPS_ && CondensedSchedule() && projectSettings_.a.keep_factors != 0 && DeferVariances() &&
               projectSettings_.a.defer_variances >= 2;
    }
    double chain_fac_budget_ = 0.0;         // HBM set aside for the chain steps' kept factors (PrepareCondensedBlocks)
    // a.reuse_factors, many small blocks (a dnasegment-default cut): the per-block steps of iterations >= 2 as ONE launch over all of them
    // (dnagpu_small_batch_*: a device-side table of the blocks' vectors, kept factors and carried junctions).  Made on first use for the set
    // of blocks at hand; SmallBatchCondense marks the blocks it served, SmallBatchSolve takes exactly those.
    dnagpu_small_batch* small_batch_ = nullptr;
    std::vector<UINT32> small_batch_blocks_;
    bool small_batch_denied_ = false, small_batch_armed_ = false;
    std::atomic<uint64_t> small_batch_steps_{0};               // block steps served that way since AdjustNetwork() began
    bool SmallBatchCondense(std::vector<UINT32>& blocks);     // serves what it can; `blocks` keeps the rest
    void SmallBatchSolve(std::vector<UINT32>& blocks);
    void SmallBatchFirstSolve(std::vector<UINT32>& blocks);
    bool EnsureSmallBatch(const std::vector<UINT32>& E);
    std::atomic<uint64_t> factor_reuses_{0}, chain_reuses_{0};    // block steps / chain steps served from a kept factor since AdjustNetwork() began
    // a chain step on the condensed block of k: elimination with the factor kept (first time) or its right-hand side through the kept factor
    void CarryCondensed(int chain, UINT32 dev_block, UINT32 block, int dir, dnagpu_matrix* W, const std::vector<UINT32>& out, dnagpu_matrix* jm);
    bool StepRhsInOneLaunch(int chain, UINT32 dev_block, UINT32 block, int dir, const dnagpu_matrix* jm_in, const std::vector<UINT32>& idx_in,
                            dnagpu_matrix* jm_out, const std::vector<UINT32>& idx_out);
public:
    uint64_t FactorReuses() const { return factor_reuses_.load(); }
    uint64_t ChainStepReuses() const { return chain_reuses_.load(); }
    uint64_t SmallBatchSteps() const { return small_batch_steps_.load(); }
    int ChainRuns() const { return lockstep_ok_ ? lock_runs_ : 0; }
private:
    void FinishDeferredVariances();
    void CarryByElimination(int chain, UINT32 dev_block, UINT32 block, dnagpu_matrix* m, const std::vector<UINT32>& out, dnagpu_matrix* jm);
    bool condensed_ok_ = false;
    // a.stage (the reference's --staged-adjustment keeps its block matrices in memory-mapped files): the rigorous variance
    // matrices live in page-locked host memory instead of HBM; switched on by itself when they would not fit
    std::map<UINT32, OscillationRecord> oscHistory_;
    bool osc_ready_ = false;
    void UpdateIterationDiagnostics();
    bool MeasurementTouchesOscillatingStation(UINT32 msrIndex) const;
    void GetMsrStations(UINT32 msrIndex, std::vector<UINT32>& out) const;
    std::string MeasurementStationNames(UINT32 msrIndex) const;
    // ---- plan mode (PlanDistributed): PrepareAdjustment's host side and memory plan without a device -- every device object the
    // prepare path would create is only counted (plan_bytes_), the free HBM is what the caller says a GPU has
    bool plan_only_ = false;
    double plan_hbm_ = 0.0, plan_bytes_ = 0.0;
    void NewMatrix(UINT32 n, dnagpu_matrix** m, UINT32 blk, const char* what);
    void NewBlock(UINT32 id, UINT32 n_stn, UINT32 n_msr, UINT32 blk, const char* what);
    void MemInfo(size_t* free_b, size_t* total_b);
    void LoadAndListNetwork();
    bool staged_ = false;
    double host_available_ = 0.0;                               // what the host could still give when the plan was made (HostMemoryAvailable)
    size_t stage_host_bytes_ = 0, stage_device_bytes_ = 0;    // the staged store's plan: packed variance matrices in host / device memory
    bool Staged() const { return staged_; }
public:
    bool IsStaged() const { return staged_; }
    void MemoryPlan(double out[12]) const;
    bool PacksItsFactor(UINT32 block) const;
private:
    std::vector<unsigned char> record_touched_;   // records whose statistics this process computed (UpdateMsrRecord)
    std::atomic<bool> chain_failed_{false};
    void OnEveryChain(const std::function<void(int)>& body);
    void ForBlocks(const std::vector<UINT32>& blocks, const std::function<void(int, UINT32)>& step);
    // a.batch_blocks: blocks of one shape as batches (dnagpu_*_batched).  phase: 0 condensing, 1 rigorous solve, 2 variance matrices
    int BatchCap() const;
    bool BatchEligible(UINT32 block, int phase) const;
    std::vector<std::vector<UINT32>> BatchGroups(const std::vector<UINT32>& blocks, int phase) const;
    void ForGroups(std::vector<std::vector<UINT32>> groups, const std::function<void(int, const std::vector<UINT32>&)>& step);
    void FitGroupsToBudget(std::vector<std::vector<UINT32>>& groups);
    void EnsurePartial(UINT32 block);
    bool BatchWorkspaces(int chain, const std::vector<UINT32>& blocks);
    void CondenseBatch(int chain, const std::vector<UINT32>& blocks);
    void NoteCondensed(UINT32 block);
    void PrepareKeptBlock(int chain, UINT32 block, int kind, dnagpu_matrix* K);
    void RigorousBatch(int chain, const std::vector<UINT32>& blocks);
    void FinishVariancesBlock(int chain, UINT32 block);
    void FinishVariancesBatch(int chain, const std::vector<UINT32>& blocks);
    std::vector<dnagpu_matrix*> kbatch_[DNAGPU_NUM_CHAINS];   // the kept blocks of the members of a batched rigorous solve
    std::atomic<uint64_t> batched_members_{0};
    double batched_flops_ = 0.0;
    double batch_budget_ = 0.0, batch_unit_ = 0.0;            // bytes left for the members' workspaces of all chains together; bytes per member
    int batch_granted_[DNAGPU_NUM_CHAINS] = {};               // members beyond the first whose workspaces chain c has been charged for
    int batch_limit_ = 0;                                     // members beyond the first that the memory budget admits (PrepareCondensedBlocks)
    void SignalExceptionAdjustment(const std::string& msg, UINT32 block);   // ADJ:10049
    void Check(int rc, UINT32 block, const char* where);
    void SetmaxCorr(double v) { maxCorr_ = v; }
    bool CombineRequired(UINT32 block) const {
        const blockMeta_t& m = v_blockMeta_[block];
        return !(m._blockLast || m._blockIsolated || m._blockFirst);
    }
    void FreeDevice();

    static inline int max_blas_threads_ = 0;
    project_settings projectSettings_;
    std::vector<station_t> bstBinaryRecords_;
    std::vector<measurement_t> bmsBinaryRecords_;
    std::vector<asl_entry_t> vAssocStnList_;
    binary_file_meta_t bst_meta_, bms_meta_;
    std::vector<std::vector<UINT32>> v_ISL_, v_JSL_, v_CML_;
    std::vector<UINT32> v_ContiguousNetList_, v_measurementCount_, v_unknownsCount_, v_parameterStationCount_;
    std::vector<blockMeta_t> v_blockMeta_;
    std::vector<std::vector<UINT32>> v_parameterStationList_;
    std::vector<std::vector<stn_appear>> v_paramStnAppearance_;
    std::vector<block_t> blocks_;
    std::vector<std::vector<double>> initial_xyz_;   // per block, for ResetAdjustment
    double* initial_dev_ = nullptr;                  // ... and once more on the device (block b at initial_off_[b])
    std::vector<size_t> initial_off_;
    dnagpu_block_table* block_table_ = nullptr;      // GNSS-only networks of many blocks: ResetAdjustment / UpdateAdjustment as one launch
    bool block_table_denied_ = false;
    void EnsureInitialOnDevice();
    bool EnsureBlockTable();

    UINT32 blockCount_ = 1;
    std::atomic<UINT32> currentBlock_{0};   // written by every chain's thread, read by the progress thread (CurrentBlock())
    UINT32 currentIteration_ = 0;
    bool isPreparing_ = false, isAdjusting_ = false, forward_ = true, isCombining_ = false;
    bool allStationsFixed_ = false, exceptionRaised_ = false;
    std::unique_ptr<DynAdjustPrinter> printer_;
    std::atomic<bool> cancel_{false};
    bool cancel_agreed_ = false;               // multi-GPU: a cancellation every rank knows of (set by AgreeOnPhase only)
    _ADJUST_STATUS_ adjustStatus_ = ADJUST_SUCCESS;
    UINT32 measurementParams_ = 0, unknownParams_ = 0, unknownsCount_ = 0;
    int degreesofFreedom_ = 0;
    double maxCorr_ = 0.0, chiSquared_ = 0.0, sigmaZero_ = 0.0, sigmaZeroSqRt_ = 0.0;
    double chiSquaredUpperLimit_ = 0.0, chiSquaredLowerLimit_ = 0.0, globalPelzerReliability_ = 0.0, criticalValue_ = 1.68;
    UINT32 potentialOutlierCount_ = 0, passFail_ = test_stat_pass;
    bool isAdjustmentQuestionable_ = false;
    bool containsNonGPS_ = false;   // MsrTally::ContainsNonGPS: the design changes with the estimates
    double var_C_ = 0.0, var_F_ = 0.0;
    std::vector<double> iterationCorrections_;
    double adjust_ms_ = 0.0;
    std::atomic<int64_t> lastBlockElapsedMs_{0};
    // DYNADJUST_PROFILE (ADJ:53-56, PrintPerformanceProfile ADJ:2562): host-side time spent issuing the formation of the normals, and
    // in the staged mode's loads / stores of the rigorous variances
    bool profileTimings_ = false;
    std::atomic<uint64_t> profileUpdateNormalsNs_{0}, profileStageLoadNs_{0}, profileStageStoreNs_{0}, stageCopiedBytes_{0}, stageWaitNs_{0};
    void PrintPerformanceProfile() const;
    void FormNormals(int chain, UINT32 block, dnagpu_matrix* W);
    std::mutex msg_mutex_;
    std::deque<UINT32> iterationQueue_;
    std::vector<double> iterationMs_;
    double solve_flops_ = 0.0;
    UINT32 solve_count_ = 0;
    double algorithmic_flops_ = 0.0;  // n^3 per inverse, the elimination's own count per dnagpu_schur_carry step
    // Of those, the flops of the MINIMAL schedule: what a GNSS-only network needs at all -- every factorisation once (iteration 1) and the
    // variance matrices once -- however many iterations run and however often a block without a kept factor makes it again.  Work
    // that is re-done counts in algorithmic_flops_ (it was executed) and not here: bench.py's roofline.frac_min_work.
    // kind: 0 = work of an iteration (minimal in iteration 1, or whenever the design moves with the estimates), 1 = done once per
    // adjustment (the variance matrices), 2 = re-done by construction (a factor made a second time)
    double min_work_flops_ = 0.0;
    void CountFlops(double f, int kind) {      // (callers hold corr_mutex_)
        algorithmic_flops_ += f;
        if (kind == 1 || (kind == 0 && (containsNonGPS_ || currentIteration_ <= 1))) min_work_flops_ += f;
    }
    UINT32 completion_count_ = 0;    // rigorous solves that completed a kept factor (a.keep_factors)
    UINT32 condense_count_ = 0;      // dnagpu_block_reduce steps (condensed schedule)
    UINT32 elimination_count_ = 0;   // of those, steps done by dnagpu_schur_carry (a.schur_carry)

    std::mutex corr_mutex_, alloc_mutex_;   // multi-thread mode: maxCorr_/solve counters, lazy allocations
    dnagpu_ctx* ctx_ = nullptr;
    dnagpu_matrix* work_[DNAGPU_NUM_CHAINS] = {};
    dnagpu_matrix* kwork_[DNAGPU_NUM_CHAINS] = {};   // the kept block of a fused rigorous solve
    dnagpu_matrix* tmpfac_[DNAGPU_NUM_CHAINS] = {};  // storage of the factor a block without a kept one makes again (TransientPartial), per chain
    bool transient_ok_ = false;                      // PrepareCondensedBlocks: such blocks exist and the conditions hold (GNSS only, light factors)
    std::atomic<uint64_t> transient_count_{0};
    std::atomic<uint64_t> unpacked_count_{0};      // rigorous solves / variance matrices that took their factor from its packed copy in HBM or host memory
    std::atomic<bool> host_factor_copies_{false};  // a condensing step has sent a packed factor to a host slot: the phase ends with the copies waited for
    dnagpu_partial* TransientPartial(int c, UINT32 k);
    bool BorrowTransientFactor(int c, UINT32 k);
    void FinishVariancesTransient(int c, UINT32 k);
    UINT32 max_unknowns_ = 0, max_junction_ = 0;
};

}  // namespace networkadjust
}  // namespace dynadjust
