// dna_adjust across GPUs: the drivers that replace dna_adjust::AdjustPhasedMultiThread (dnaadjust-multi.cpp:92-244) when the
// blocks of one network are spread over several MI355X -- one rank per GPU, the exchange step over RCCL on xGMI (dist_comm.hpp).
//
// Condensed schedule (a.schur_carry), per iteration:
//   (A) CondenseBlocks(own blocks)          every block reduced to its shared stations on its owner's GPU, no dependency
//   exchange                                 ncclBroadcast of each condensed block (matrix + reduced right-hand side, in place)
//   (B) CondensedChains()                    the reference's forward / reverse chains (AdjustPhasedForward ADJ:2756,
//                                            AdjustPhasedReverseCombine ADJ:3461) on the condensed blocks, on every rank
//   (C) RigorousBlocks(own blocks)           the one full inverse per block, on its owner's GPU
//   sync                                     ncclAllReduce(sum) of the coordinate vector (owner writes, others zero) with the
//                                            ranks' largest corrections appended: the convergence test of ADJ:2639 on identical
//                                            data everywhere
// Reference schedule (a.schur_carry = 0): rank 0 runs the forward chain while rank 1 runs the reverse chain
// (adjust_forward_thread / adjust_reverse_thread, dnaadjust-multi.cpp:365 / 475); the junction matrices the combination solves
// need (v_junctionVariancesFwd_[k-1], v_junctionVariances_[k] and their estimates) travel point-to-point (ncclSend / ncclRecv,
// one group); the combination solves (combine thread, dnaadjust-multi.cpp:593) go round-robin over all ranks.
//
// One process per GPU, or one process with one host thread per GPU (a.devices): the same code, rank by rank.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <exception>
#include <numeric>
#include <set>
#include <thread>

#include "dna_adjust.hpp"

namespace dynadjust {
namespace networkadjust {

namespace {
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// Who holds which block: identical on every rank (pure function of the segmentation).
//   condensed schedule: contiguous runs of blocks, the largest run's sum of n^3 as small as possible (the blocks of a rank are
//     neighbours: what the two-level chain needs, and as balanced as any other assignment when blocks are of similar size)
//   reference schedule: last / isolated blocks -> the forward rank 0 (their forward solution is rigorous, ADJ:3033), first blocks
//     -> the reverse rank 1, intermediate blocks -> their combination solve's rank (round-robin)
void dna_adjust::ComputeBlockOwners(bool condensed) {
    const int W = std::max(1, projectSettings_.a.dist_world);
    owner_.assign(blockCount_, 0);
    if (W <= 1 || projectSettings_.a.adjust_mode == SimultaneousMode) return;
    if (!condensed) {
        int idx = 0;
        for (UINT32 k = 0; k < blockCount_; ++k) {
            const blockMeta_t& m = v_blockMeta_[k];
            if (m._blockLast || m._blockIsolated)
                owner_[k] = 0;
            else if (m._blockFirst)
                owner_[k] = W > 1 ? 1 : 0;
            else
                owner_[k] = idx++ % W;
        }
        return;
    }
    std::vector<double> cost(blockCount_);
    double total = 0.0, largest = 0.0;
    for (UINT32 k = 0; k < blockCount_; ++k) {
        const double n = 3.0 * (double)v_parameterStationList_[k].size();
        cost[k] = n * n * n;
        total += cost[k];
        largest = std::max(largest, cost[k]);
    }
    auto parts_needed = [&](double cap) {
        int parts = 1;
        double load = 0.0;
        for (UINT32 k = 0; k < blockCount_; ++k) {
            if (load + cost[k] > cap && load > 0.0) {
                ++parts;
                load = 0.0;
            }
            load += cost[k];
        }
        return parts;
    };
    double lo = largest, hi = total;
    for (int it = 0; it < 80; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (parts_needed(mid) <= W)
            hi = mid;
        else
            lo = mid;
    }
    const double cap = hi * (1.0 + 1e-12);
    int r = 0;
    double load = 0.0;
    for (UINT32 k = 0; k < blockCount_; ++k) {
        if (load + cost[k] > cap && load > 0.0 && r + 1 < W) {
            ++r;
            load = 0.0;
        }
        // (never leave the ranks behind without a block while blocks remain)
        if ((int)(blockCount_ - k) <= W - 1 - r && load > 0.0 && r + 1 < W) {
            ++r;
            load = 0.0;
        }
        owner_[k] = r;
        load += cost[k];
    }
}

double* dna_adjust::ExchangeBuffer(size_t doubles) {
    if (xbuf_cap_ < doubles) {
        if (xbuf_dev_) dnagpu_device_free(ctx_, xbuf_dev_);
        xbuf_dev_ = nullptr;
        xbuf_cap_ = 0;
        void* p = nullptr;
        Check(dnagpu_device_alloc(ctx_, doubles * sizeof(double), &p), 0, "exchange buffer");
        xbuf_dev_ = (double*)p;
        xbuf_cap_ = doubles;
    }
    return xbuf_dev_;
}

// body() on this rank; afterwards all ranks learn whether any of them failed, so that nobody waits in the next collective for a
// rank that has left (the reference's threads do the same through their shared exception pointers, dnaadjust-multi.cpp:182-190)
void dna_adjust::AgreeOnPhase(const char* phase, const std::function<void()>& body) {
    std::exception_ptr mine;
    try {
        body();
    } catch (...) {
        mine = std::current_exception();
    }
    double others = 0.0;
    try {
        double* flag = ExchangeBuffer(1);
        double v = mine ? 1.0 : 0.0;
        Check(dnagpu_copy(ctx_, flag, &v, sizeof(double)), 0, "exchange");
        comm_->all_reduce_sum(flag, 1);
        comm_->wait();
        Check(dnagpu_copy(ctx_, &v, flag, sizeof(double)), 0, "exchange");
        others = v - (mine ? 1.0 : 0.0);
    } catch (...) {
        if (!mine) throw;
    }
    if (mine) std::rethrow_exception(mine);
    if (others > 0.5) SignalExceptionAdjustment(std::string("AdjustNetwork(): the adjustment failed on another GPU (") + phase + ").", currentBlock_);
}

// every condensed block travels from its owner to every rank: matrix and reduced right-hand side, in place, one group
void dna_adjust::ExchangeCondensed() {
    const double t0 = wall_ms();
    comm_->group_begin();
    for (UINT32 k = 0; k < blockCount_; ++k) {
        block_t& B = blocks_[k];
        if (!B.red) continue;
        Check(dnagpu_matrix_resize(ctx_, B.red, (UINT32)B.keep.size() * 3), k, "exchange");
        double *F = nullptr, *v = nullptr;
        UINT32 np = 0;
        dnagpu_matrix_device_pointers(B.red, &F, &v, &np);
        comm_->broadcast(F, (size_t)np * np, BlockOwner(k));
        comm_->broadcast(v, np, BlockOwner(k));
    }
    comm_->group_end();
    comm_->wait();
    exchange_ms_ += wall_ms() - t0;
}

// rigorous coordinates of every block and the largest correction, on every rank
void dna_adjust::SyncCoordinates() {
    const double t0 = wall_ms();
    const int W = DistWorld(), me = DistRank();
    std::vector<size_t> off(blockCount_ + 1, 0);
    for (UINT32 k = 0; k < blockCount_; ++k) off[k + 1] = off[k] + 3 * v_parameterStationList_[k].size();
    const size_t total = off[blockCount_] + (size_t)W;
    std::vector<double> flat(total, 0.0), bx;
    for (UINT32 k = 0; k < blockCount_; ++k)
        if (OwnsBlock(k)) {
            GetBlockStations(k, 2, bx);
            std::copy(bx.begin(), bx.end(), flat.begin() + off[k]);
        }
    flat[off[blockCount_] + me] = maxCorr_;
    double* dev = ExchangeBuffer(total);
    Check(dnagpu_copy(ctx_, dev, flat.data(), total * sizeof(double)), 0, "exchange");
    comm_->all_reduce_sum(dev, total);
    comm_->wait();
    Check(dnagpu_copy(ctx_, flat.data(), dev, total * sizeof(double)), 0, "exchange");
    for (UINT32 k = 0; k < blockCount_; ++k)
        if (!OwnsBlock(k)) SetBlockStationsAll(k, flat.data() + off[k]);
    for (int r = 0; r < W; ++r) PhasedNoteCorrection(flat[off[blockCount_] + r]);
    exchange_ms_ += wall_ms() - t0;
}

void dna_adjust::DistributedCondensedIteration() {
    std::vector<UINT32> mine;
    for (UINT32 k = 0; k < blockCount_; ++k)
        if (OwnsBlock(k)) mine.push_back(k);
    AgreeOnPhase("condensing the blocks", [&] { CondenseBlocks(mine); });
    if (IsCancelled()) return;
    ExchangeCondensed();
    const double t0 = wall_ms();
    AgreeOnPhase("junction chains", [&] { CondensedChains(); });
    chain_ms_ += wall_ms() - t0;
    if (IsCancelled()) return;
    AgreeOnPhase("rigorous block solutions", [&] { RigorousBlocks(mine); });
}

namespace {
struct junction_msg {
    int kind;   // 0 forward (jfwd of `block`), 1 reverse (jrev of `block`)
    UINT32 block;
    int src, dst;
};
}  // namespace

void dna_adjust::DistributedReferenceIteration() {
    const int W = DistWorld(), me = DistRank();
    const int fwd_rank = 0, rev_rank = W > 1 ? 1 : 0;
    AgreeOnPhase("forward and reverse passes", [&] {
        if (me == fwd_rank) {
            forward_ = true;
            for (UINT32 k = 0; k < blockCount_ && !IsCancelled(); ++k) {
                currentBlock_ = k;
                PhasedForwardBlock(0, k);        // notes the correction of a last / isolated block itself
            }
        }
        if (me == rev_rank) {
            forward_ = false;
            const int c = (me == fwd_rank && NumChains() > 1) ? 1 : 0;
            for (UINT32 kk = blockCount_; kk-- > 0 && !IsCancelled();) {
                const blockMeta_t& m = v_blockMeta_[kk];
                if (m._blockIsolated) continue;
                currentBlock_ = kk;
                const double mv = PhasedReverseBlock(c, kk);
                if (m._blockFirst && !m._blockLast) {     // first block of a network: rigorous now
                    PhasedNoteCorrection(mv);
                    PhasedFinaliseBlock(c, kk);
                }
            }
        }
    });
    if (IsCancelled()) return;
    // the junction payloads of the combination solves: jfwd[k-1] from the forward rank, jrev[k] from the reverse rank
    const double t0 = wall_ms();
    std::vector<junction_msg> msgs;
    for (UINT32 k = 0; k < blockCount_; ++k) {
        if (!CombineRequired(k)) continue;
        const int o = BlockOwner(k);
        if (fwd_rank != o) msgs.push_back({0, k - 1, fwd_rank, o});
        if (rev_rank != o) msgs.push_back({1, k, rev_rank, o});
    }
    comm_->group_begin();
    for (const junction_msg& g : msgs) {
        if (g.src != me && g.dst != me) continue;
        dnagpu_matrix* jm = g.kind == 0 ? blocks_[g.block].jfwd : blocks_[g.block].jrev;
        if (!jm) continue;
        if (g.dst == me) Check(dnagpu_matrix_resize(ctx_, jm, JunctionUnknowns(g.block)), g.block, "exchange");
        double *F = nullptr, *v = nullptr;
        UINT32 np = 0;
        dnagpu_matrix_device_pointers(jm, &F, &v, &np);
        if (g.src == me) {
            comm_->send(F, (size_t)np * np, g.dst);
            comm_->send(v, np, g.dst);
        } else {
            comm_->recv(F, (size_t)np * np, g.src);
            comm_->recv(v, np, g.src);
        }
    }
    comm_->group_end();
    comm_->wait();
    exchange_ms_ += wall_ms() - t0;
    AgreeOnPhase("combination solutions", [&] {
        isCombining_ = true;
        for (UINT32 k = 0; k < blockCount_ && !IsCancelled(); ++k) {
            if (!CombineRequired(k) || !OwnsBlock(k)) continue;
            currentBlock_ = k;
            const double mv = PhasedCombineBlock(0, k);
            PhasedNoteCorrection(mv);
            PhasedFinaliseBlock(0, k);
        }
        isCombining_ = false;
    });
}

// AdjustPhased (ADJ:2579-2670) across the ranks
void dna_adjust::AdjustPhasedDistributed() {
    currentIteration_ = 0;
    for (UINT32 i = 0; i < projectSettings_.a.max_iterations; ++i) {
        if (IsCancelled()) break;
        const double it_t0 = wall_ms();
        PhasedBeginIteration();
        if (CondensedSchedule())
            DistributedCondensedIteration();
        else
            DistributedReferenceIteration();
        if (IsCancelled()) break;
        Check(dnagpu_sync(ctx_), 0, "AdjustNetwork()");
        SyncCoordinates();
        NoteIterationDone(it_t0);
        if (!PhasedEndIteration()) break;
    }
    PhasedFinish();
}

// GenerateStatistics (ADJ:6802) when the rigorous variances are spread over the ranks: every rank computes the precisions of
// the adjusted measurements, the per-record statistics and the chi-square terms of its own blocks; one all-reduce(sum) of
// (chi-square, outlier count, 9 doubles per .bms record -- zero where not computed here) gives every rank the whole picture
void dna_adjust::GenerateStatisticsDistributed() {
    AgreeOnPhase("statistics", [&] {
        UpdateAdjustment(false);
        StatisticsBegin();
        for (UINT32 k = 0; k < blockCount_; ++k)
            if (OwnsBlock(k)) StatisticsBlock(k);
    });
    const size_t n = bmsBinaryRecords_.size();
    std::vector<double> host(2 + 9 * n, 0.0);
    host[0] = chiSquared_;
    host[1] = (double)potentialOutlierCount_;
    GetRecordStatistics(host.data() + 2);
    double* dev = ExchangeBuffer(host.size());
    Check(dnagpu_copy(ctx_, dev, host.data(), host.size() * sizeof(double)), 0, "exchange");
    comm_->all_reduce_sum(dev, host.size());
    comm_->wait();
    Check(dnagpu_copy(ctx_, host.data(), dev, host.size() * sizeof(double)), 0, "exchange");
    SetPartials(host[0], (UINT32)std::llround(host[1]));
    SetRecordStatistics(host.data() + 2);
    StatisticsFinish();
}

// Rigorous variances (packed) and adjusted-measurement precisions of block b on rank 0, wherever they were computed.
// Collective over the processes; the vectors stay empty on every rank but 0.
void dna_adjust::CollectBlockResults(UINT32 b, std::vector<double>& packed, std::vector<double>& prec) {
    const int me = DistRank(), o = BlockOwner(b);
    const size_t n = 3 * v_parameterStationList_[b].size(), cnt = n * (n + 1) / 2;
    const size_t rows = 6 * blocks_[b].stn1.size() + blocks_[b].t_type.size();
    packed.clear();
    prec.clear();
    if (o == 0) {
        if (me == 0) {
            GetBlockRigorousVariancesPacked(b, packed);
            prec = blocks_[b].prec_adj_msrs;
        }
        return;
    }
    if (me != 0 && me != o) return;
    double* dev = ExchangeBuffer(cnt + rows + 1);
    if (me == o) {
        GetBlockRigorousVariancesPacked(b, packed);
        std::vector<double> tail(rows + 1, 0.0);
        tail[0] = blocks_[b].prec_adj_msrs.size() == rows ? 1.0 : 0.0;
        if (tail[0] > 0.5) std::copy(blocks_[b].prec_adj_msrs.begin(), blocks_[b].prec_adj_msrs.end(), tail.begin() + 1);
        Check(dnagpu_copy(ctx_, dev, packed.data(), cnt * sizeof(double)), b, "exchange");
        Check(dnagpu_copy(ctx_, dev + cnt, tail.data(), tail.size() * sizeof(double)), b, "exchange");
        comm_->send(dev, cnt + rows + 1, 0);
        comm_->wait();
        packed.clear();
    } else {
        comm_->recv(dev, cnt + rows + 1, o);
        comm_->wait();
        packed.resize(cnt);
        std::vector<double> tail(rows + 1);
        Check(dnagpu_copy(ctx_, packed.data(), dev, cnt * sizeof(double)), b, "exchange");
        Check(dnagpu_copy(ctx_, tail.data(), dev + cnt, tail.size() * sizeof(double)), b, "exchange");
        if (tail[0] > 0.5) prec.assign(tail.begin() + 1, tail.end());
    }
}

// ---- one process, several GPUs ------------------------------------------------------------------------------------------------
// a.devices = {d0, d1, ...}: this instance becomes rank 0 on d0 and creates one more instance per further GPU; every entry point
// that is collective runs on all of them, one host thread each (the reference's --multi-thread starts its forward, reverse and
// combination threads inside AdjustPhasedMultiThread the same way).  Transport: RCCL when every rank has a GPU of its own,
// device-to-device copies when ranks share one.
void dna_adjust::OnEveryDevice(const std::function<void(dna_adjust&)>& body) {
    const int N = DeviceInstances();
    std::vector<std::exception_ptr> errors(N);
    auto run = [&](int r) {
        dna_adjust* a = DeviceInstance(r);
        a->in_collective_ = true;
        try {
            body(*a);
        } catch (...) {
            errors[r] = std::current_exception();
        }
        a->in_collective_ = false;
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < N; ++r) threads.emplace_back(run, r);
    run(0);
    for (std::thread& t : threads) t.join();
    // this instance's own failure first (it carries the block number the caller will report), else the first peer's
    for (int r = 0; r < N; ++r)
        if (errors[r]) {
            adjustStatus_ = ADJUST_EXCEPTION_RAISED;
            exceptionRaised_ = true;
            std::rethrow_exception(errors[r]);
        }
}

void dna_adjust::PrepareMultiDevice(const project_settings& p) {
    peers_.clear();
    FreeDevice();
    comm_.reset();
    const std::vector<int> devs = p.a.devices;
    const int N = (int)devs.size();
    bool distinct = std::set<int>(devs.begin(), devs.end()).size() == devs.size();
    std::string transport = p.a.dist_transport;
    if (transport.empty()) transport = (distinct && rccl_available()) ? "rccl" : "local";
    if (transport == "rccl" && !distinct)
        SignalExceptionAdjustment("PrepareAdjustment(): RCCL needs a GPU per rank; a.devices names one twice (use a.dist_transport = \"local\").", 0);
    if (transport != "rccl" && transport != "local") SignalExceptionAdjustment("PrepareAdjustment(): unknown a.dist_transport '" + transport + "'.", 0);
    for (int r = 1; r < N; ++r) {
        peers_.emplace_back(new dna_adjust());
        peers_.back()->is_peer_ = true;
    }
    std::vector<std::shared_ptr<DistComm>> local;
    unsigned char id[DIST_UNIQUE_ID_BYTES] = {0};
    try {
        if (transport == "local")
            local = local_comm_create(N, devs);
        else
            rccl_unique_id(id);
    } catch (const std::exception& e) {
        peers_.clear();
        SignalExceptionAdjustment(std::string("PrepareAdjustment(): ") + e.what(), 0);
    }
    try {
        // (OnEveryDevice hands out the instances; each looks its rank up by identity)
        OnEveryDevice([&](dna_adjust& a) {
            int r = 0;
            for (int q = 0; q < N; ++q)
                if (DeviceInstance(q) == &a) r = q;
            project_settings ps = p;
            ps.a.devices.clear();
            ps.a.device = devs[r];
            ps.a.dist_rank = r;
            ps.a.dist_world = N;
            a.comm_ = transport == "local" ? local[r] : rccl_comm_create(r, N, id, devs[r]);
            a.PrepareAdjustment(ps);
        });
    } catch (...) {
        peers_.clear();
        throw;
    }
    projectSettings_.a.devices = devs;     // (what the caller asked for)
}

}  // namespace networkadjust
}  // namespace dynadjust
