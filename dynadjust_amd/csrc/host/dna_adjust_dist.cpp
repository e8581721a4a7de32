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
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <exception>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <sstream>
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

// PrepareAdjustment's plan for every rank of `world` GPUs with `hbm_bytes` each, without a device (plan_only_: device objects are counted,
// not made).  The schedule functions are the ones the adjustment runs -- ComputeBlockOwners, DecideStaging, PrepareCondensedBlocks,
// PrepareTwoLevel -- so what this prints is what PrepareAdjustment would decide, and tests/test_dist_plan.py pins it on the CPU.
std::string dna_adjust::PlanDistributed(const project_settings& projectSettings, int world, double hbm_bytes) {
    FreeDevice();
    peers_.clear();
    if (world < 1) world = 1;
    plan_only_ = true;
    plan_hbm_ = hbm_bytes;
    const std::shared_ptr<DistComm> attached = comm_;      // (a communicator the host attached survives the plan)
    std::ostringstream js;
    js.precision(12);
    try {
        projectSettings_ = projectSettings;
        projectSettings_.a.dist_world = world;
        projectSettings_.a.devices.clear();
        var_C_ = projectSettings_.a.fixed_std_dev * projectSettings_.a.fixed_std_dev;
        var_F_ = projectSettings_.a.free_std_dev * projectSettings_.a.free_std_dev;
        LoadAndListNetwork();
        auto sq = [](double n) { return (n + 256.0) * (n + 256.0) * 8.0; };
        auto padsq = [](double n) { const double p = std::ceil(n / 128.0) * 128.0; return (p * p + p) * 8.0; };
        double total_n3 = 0.0;
        for (UINT32 k = 0; k < blockCount_; ++k) total_n3 += std::pow(3.0 * (double)v_parameterStationList_[k].size(), 3.0);
        js << "{\"world\": " << world << ", \"blocks\": " << blockCount_ << ", \"stations\": " << bstBinaryRecords_.size() << ", \"hbm_bytes_per_gpu\": " << hbm_bytes
           << ", \"ranks\": [";
        for (int r = 0; r < world; ++r) {
            comm_ = plan_comm_create(r, world);
            projectSettings_.a.dist_rank = r;
            mt_chains_ = DNAGPU_DEFAULT_CHAINS;
            if (const char* e = getenv("DNAGPU_CHAINS")) mt_chains_ = std::max(2, std::min(DNAGPU_NUM_CHAINS, atoi(e)));
            staged_ = projectSettings_.a.stage != 0;
            plan_bytes_ = 0.0;
            PrepareBlocks();
            const double prepared = plan_bytes_;
            int first = -1, last = -1, own = 0, kept = 0;
            double n3 = 0.0, rig = 0.0, kept_bytes = 0.0;
            for (UINT32 k = 0; k < blockCount_; ++k) {
                if (!OwnsBlock(k)) continue;
                if (first < 0) first = (int)k;
                last = (int)k;
                ++own;
                const double n = 3.0 * (double)v_parameterStationList_[k].size();
                n3 += n * n * n;
                rig += sq(n);
                if (blocks_[k].part_allowed) {
                    ++kept;
                    if (!blocks_[k].part_in_rigvar) kept_bytes += sq(n);
                }
            }
            const double chains_ws = 3.0 * (double)NumChains() * sq((double)max_unknowns_);
            const double variances_hbm = staged_ ? (double)stage_device_bytes_ : rig;
            const double batch_ws = (double)std::min(batch_limit_, (DNAGPU_BATCH_MAX - 1) * NumChains()) * batch_unit_;
            const double committed = prepared + chains_ws + variances_hbm + kept_bytes + (transient_ok_ ? (double)NumChains() * sq((double)max_unknowns_) : 0.0);
            // the exchanges of one iteration, bytes this rank takes part in
            double exch_blocks = 0.0, exch_runs = 0.0;
            for (UINT32 k = 0; k < blockCount_; ++k) exch_blocks += (double)CondensedPayloadDoubles(k) * 8.0;
            if (two_level_ok_)
                for (const segment_t& g : segs_) exch_runs += padsq(3.0 * (double)g.stations.size());
            double coords = (double)world * 8.0;
            for (UINT32 k = 0; k < blockCount_; ++k) coords += 24.0 * (double)v_parameterStationList_[k].size();
            js << (r ? ", " : "") << "{\"rank\": " << r << ", \"first_block\": " << first << ", \"last_block\": " << last << ", \"own_blocks\": " << own
               << ", \"share_of_sum_n3\": " << (total_n3 > 0 ? n3 / total_n3 : 0.0) << ", \"chains\": " << NumChains() << ", \"condensed_schedule\": "
               << (condensed_ok_ ? "true" : "false") << ", \"two_level_chains\": " << (two_level_ok_ ? "true" : "false") << ", \"staged\": " << (staged_ ? "true" : "false")
               << ", \"hbm\": {\"blocks_and_chain_data\": " << prepared << ", \"chain_workspaces\": " << chains_ws << ", \"variance_matrices\": " << variances_hbm
               << ", \"staged_in_host_memory\": " << (double)stage_host_bytes_ << ", \"kept_factors_own_storage\": " << kept_bytes << ", \"batch_workspaces_up_to\": " << batch_ws
               << ", \"committed\": " << committed << ", \"fits\": " << (committed <= hbm_bytes - 1.5e9 ? "true" : "false") << "}"
               << ", \"blocks_keeping_their_factor\": " << kept << ", \"factors_made_again\": " << (transient_ok_ ? "true" : "false") << ", \"batch_members_beyond_first\": "
               << batch_limit_ << ", \"exchange_bytes_per_iteration\": {\"condensed_blocks_one_level\": " << exch_blocks << ", \"run_systems_two_level\": " << exch_runs
               << ", \"coordinates_all_reduce\": " << coords << "}";
            if (two_level_ok_) {
                const segment_t& g = segs_[r];
                js << ", \"run\": {\"a\": " << g.a << ", \"b\": " << g.b << ", \"end_stations\": " << g.stations.size() << ", \"towards_previous\": " << g.posL.size()
                   << ", \"towards_next\": " << g.posR.size() << ", \"merges\": [";
                for (size_t i = 0; i < g.steps.size(); ++i)
                    js << (i ? ", " : "") << "{\"block\": " << (g.a + 1 + i) << ", \"stations\": " << g.steps[i].n_stn << ", \"stay\": " << g.steps[i].keep.size() << "}";
                js << "], \"level2_steps_each_way\": " << (world - 1) << ", \"level3_steps_each_way\": " << (g.b - g.a) << "}";
            }
            js << ", \"owners\": [";
            for (UINT32 k = 0; k < blockCount_; ++k) js << (k ? "," : "") << BlockOwner(k);
            js << "]}";
            FreeTwoLevel();
        }
        js << "]}";
    } catch (...) {
        comm_ = attached;
        plan_only_ = false;
        blocks_.clear();
        throw;
    }
    comm_ = attached;
    plan_only_ = false;
    blocks_.clear();
    blockCount_ = 0;
    return js.str();
}

// the exchange of the intra-block distributed inverse (dnagpu_set_inverse_exchange): no exception may cross the C boundary
int dna_adjust::ExchangeTrampoline(void* self, void* stream, int nparts, double* const* bufs, const size_t* counts) {
    dna_adjust* a = static_cast<dna_adjust*>(self);
    try {
        a->comm_->broadcast_parts_on((hipStream_t)stream, nparts, bufs, counts);
        return 0;
    } catch (...) {
        return -1;
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
// test hook (dnaadj_debug_stall_rank): rank `g_stall_rank` sleeps `g_stall_seconds` before its n-th agreement -- a rank that hangs
// in a kernel, as far as the others can tell
static std::atomic<int> g_stall_rank{-1};
static std::atomic<long> g_stall_countdown{0};
static std::atomic<double> g_stall_seconds{0.0};
void debug_stall_rank(int rank, long nth_agreement, double seconds) {
    g_stall_rank.store(rank);
    g_stall_countdown.store(nth_agreement);
    g_stall_seconds.store(seconds);
}

void dna_adjust::AgreeOnPhase(const char* phase, const std::function<void()>& body) {
    std::exception_ptr mine;
    if (g_stall_rank.load() == DistRank() && g_stall_countdown.load() > 0 && g_stall_countdown.fetch_sub(1) == 1) {
        g_stall_rank.store(-1);
        std::this_thread::sleep_for(std::chrono::duration<double>(g_stall_seconds.load()));
    }
    try {
        body();
    } catch (const std::exception& e) {
        mine = std::current_exception();
    } catch (...) {
        mine = std::current_exception();
    }
    double others = 0.0, cancelled = 0.0;
    try {
        // two words: "I failed" and "I was cancelled" -- the sums tell every rank the same thing at the same point of the schedule
        if (!agree_dev_) {
            void* q = nullptr;
            Check(dnagpu_device_alloc(ctx_, 2 * sizeof(double), &q), 0, "exchange");
            agree_dev_ = (double*)q;
        }
        double* flag = agree_dev_;
        double v[2] = {mine ? 1.0 : 0.0, IsCancelled() ? 1.0 : 0.0};
        Check(dnagpu_copy(ctx_, flag, v, sizeof(v)), 0, "exchange");
        comm_->all_reduce_sum(flag, 2);
        comm_->wait();
        Check(dnagpu_copy(ctx_, v, flag, sizeof(v)), 0, "exchange");
        others = v[0] - (mine ? 1.0 : 0.0);
        cancelled = v[1];
    } catch (...) {
        if (!mine) throw;
    }
    if (cancelled > 0.5) {
        // CancelAdjustment() reached (at least) one rank: from here on every rank is cancelled, and -- unlike cancel_, which a signal
        // handler may set on one rank at any moment -- cancel_agreed_ changes at agreements only, so the ranks leave the loops together
        cancel_.store(true);
        cancel_agreed_ = true;
    }
    if (mine) std::rethrow_exception(mine);
    if (others > 0.5) SignalExceptionAdjustment(std::string("AdjustNetwork(): the adjustment failed on another GPU (") + phase + ").", currentBlock_);
}

// every condensed block travels from its owner to every rank: matrix and reduced right-hand side, in place, one group
void dna_adjust::ExchangeCondensed() {
    const double t0 = wall_ms();
    // everything that can fail locally (device allocations inside resize) happens and is agreed on BEFORE a collective is posted:
    // a rank that threw between group_begin and group_end would leave the others in ncclBroadcast for good
    struct part_t { double *F, *v; UINT32 np; int root; };
    std::vector<part_t> parts;
    AgreeOnPhase("exchange of the condensed blocks (preparation)", [&] {
        for (UINT32 k = 0; k < blockCount_; ++k) {
            block_t& B = blocks_[k];
            if (!B.red) continue;
            Check(dnagpu_matrix_resize(ctx_, B.red, (UINT32)B.keep.size() * 3), k, "exchange");
            part_t q{nullptr, nullptr, 0, BlockOwner(k)};
            dnagpu_matrix_device_pointers(B.red, &q.F, &q.v, &q.np);
            parts.push_back(q);
        }
    });
    comm_->group_begin();
    for (const part_t& q : parts) {
        comm_->broadcast(q.F, (size_t)q.np * q.np, q.root);
        comm_->broadcast(q.v, q.np, q.root);
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
    double* dev = nullptr;
    AgreeOnPhase("coordinates (preparation)", [&] {
        for (UINT32 k = 0; k < blockCount_; ++k)
            if (OwnsBlock(k)) {
                GetBlockStations(k, 2, bx);
                std::copy(bx.begin(), bx.end(), flat.begin() + off[k]);
            }
        flat[off[blockCount_] + me] = maxCorr_;
        dev = ExchangeBuffer(total);
        Check(dnagpu_copy(ctx_, dev, flat.data(), total * sizeof(double)), 0, "exchange");
    });
    comm_->all_reduce_sum(dev, total);
    comm_->wait();
    AgreeOnPhase("coordinates", [&] {
        Check(dnagpu_copy(ctx_, flat.data(), dev, total * sizeof(double)), 0, "exchange");
        for (UINT32 k = 0; k < blockCount_; ++k)
            if (!OwnsBlock(k)) SetBlockStationsAll(k, flat.data() + off[k]);
    });
    for (int r = 0; r < W; ++r) PhasedNoteCorrection(flat[off[blockCount_] + r]);
    exchange_ms_ += wall_ms() - t0;
}

// ---- two-level condensed chains ------------------------------------------------------------------------------------------------
// With every rank running both chains on all condensed blocks, the chain phase is 2 (B - 1) sequential steps everywhere and every
// condensed block travels to every rank (cfg4: 128 blocks x 288 MB per iteration).  Ranks own contiguous runs of blocks, so:
//   level 1  every rank condenses its own run once more, to the junction stations of the run's two ends: the run's condensed blocks
//            are merged one after the other and the stations no longer needed are eliminated (dnagpu_block_reduce)
//   exchange one such system per rank (not per block) is broadcast
//   level 2  every rank runs the forward and the reverse chain over the W run systems: W - 1 steps each way; they leave the junction
//            weights / estimates at every run boundary -- exactly what the block-level chains carry across it
//   level 3  every rank runs both chains over its own blocks only, from the boundary values of level 2
// Depth B / W + W + B / W instead of B; the same additions in the same order inside every run, so the results agree with the
// one-level chains to rounding.
void dna_adjust::FreeTwoLevel() {
    for (segment_t& g : segs_) {
        if (g.S) dnagpu_matrix_destroy(ctx_, g.S);
        for (dnagpu_matrix*& m : g.M)
            if (m) dnagpu_matrix_destroy(ctx_, m);
    }
    segs_.clear();
    two_level_ok_ = false;
}

void dna_adjust::PrepareTwoLevel() {
    FreeTwoLevel();
    const int W = DistWorld(), me = DistRank();
    // (every condition below is the same on every rank: nothing that depends on a rank's free memory)
    if (!CondensedSchedule() || projectSettings_.a.reuse_inverses != 0) return;
    // (the same three levels with the runs as virtual ranks of ONE GPU -- every run on a chain of its own -- were measured in round 4 and
    //  again in round 5's pruning note: eight streams of dependent small kernels slow each other down; removed, profiles/HISTORY.md)
    if (W < 2 || !projectSettings_.a.dist_two_level) return;
    // one contiguous network, every rank a run of at least one block
    if (!v_blockMeta_[0]._blockFirst || !v_blockMeta_[blockCount_ - 1]._blockLast) return;
    for (UINT32 k = 0; k < blockCount_; ++k) {
        const blockMeta_t& m = v_blockMeta_[k];
        if (m._blockIsolated || (m._blockFirst && k != 0) || (m._blockLast && k != blockCount_ - 1)) return;
        if (blocks_[k].keep.empty()) return;
    }
    std::vector<int> run_of(blockCount_, 0);
    for (UINT32 k = 0; k < blockCount_; ++k) run_of[k] = BlockOwner(k);
    std::vector<int> first(W, -1), last(W, -1);
    for (UINT32 k = 0; k < blockCount_; ++k) {
        const int r = run_of[k];
        if (first[r] < 0) first[r] = (int)k;
        if (last[r] >= 0 && last[r] != (int)k - 1) return;       // runs must be contiguous
        last[r] = (int)k;
    }
    for (int r = 0; r < W; ++r)
        if (first[r] < 0) return;
    auto gid = [&](UINT32 k, UINT32 keep_pos) { return v_parameterStationList_[k][blocks_[k].keep[keep_pos]]; };
    auto position = [](const std::vector<UINT32>& sorted, UINT32 g) {
        auto it = std::lower_bound(sorted.begin(), sorted.end(), g);
        return (it != sorted.end() && *it == g) ? (long)(it - sorted.begin()) : -1L;
    };
    segs_.assign(W, segment_t());
    for (int r = 0; r < W; ++r) {
        segment_t& g = segs_[r];
        g.a = (UINT32)first[r];
        g.b = (UINT32)last[r];
        const block_t& A = blocks_[g.a];
        const block_t& Bk = blocks_[g.b];
        std::vector<UINT32> L, R;
        for (UINT32 p : A.c_prev) L.push_back(gid(g.a, p));
        if (g.b + 1 < blockCount_)
            for (UINT32 p : Bk.c_next) R.push_back(gid(g.b, p));
        if ((r > 0 && L.empty()) || (r + 1 < W && R.empty())) {
            segs_.clear();
            return;
        }
        g.stations = L;
        g.stations.insert(g.stations.end(), R.begin(), R.end());
        std::sort(g.stations.begin(), g.stations.end());
        g.stations.erase(std::unique(g.stations.begin(), g.stations.end()), g.stations.end());
        for (UINT32 s : L) g.posL.push_back((UINT32)position(g.stations, s));
        for (UINT32 s : R) g.posR.push_back((UINT32)position(g.stations, s));
        std::set<UINT32> inL(L.begin(), L.end());
        for (UINT32 q = 0; q < g.stations.size(); ++q) {
            const UINT32 s = g.stations[q];
            if (inL.count(s)) {
                g.dstA.push_back(q);
                g.srcA.push_back(LocalIndex(g.a, s));
            } else {
                g.dstB.push_back(q);
                g.srcB.push_back(LocalIndex(g.b, s));
            }
        }
        for (UINT32 k = g.a; k <= g.b; ++k) {
            auto pick = [&](const constraint_list& src, constraint_list& dst) {
                for (size_t i = 0; i < src.stn.size(); ++i) {
                    const long q = position(g.stations, gid(k, src.stn[i]));
                    if (q < 0) continue;
                    dst.stn.push_back((UINT32)q);
                    dst.w9.insert(dst.w9.end(), src.w9.begin() + 9 * i, src.w9.begin() + 9 * i + 9);
                }
            };
            pick(blocks_[k].ccon_fwd, g.con_fwd);
            pick(blocks_[k].ccon_rev, g.con_rev);
        }
        g.dev_block = 3 * blockCount_ + (UINT32)r;
        NewBlock(g.dev_block, (UINT32)g.stations.size(), 0, g.a, "PrepareAdjustment(): run system");
        NewMatrix((UINT32)g.stations.size() * 3, &g.S, g.a, "PrepareAdjustment(): run system");
    }
    // the merges of the own run
    for (int rr = 0; rr < W; ++rr) {
        if (rr != me) continue;
        segment_t& g = segs_[rr];
        std::vector<UINT32> prev;
        for (UINT32 p = 0; p < blocks_[g.a].keep.size(); ++p) prev.push_back(gid(g.a, p));
        size_t max_keep = 0;
        for (UINT32 k = g.a + 1; k <= g.b; ++k) {
            seg_step_t st;
            std::vector<UINT32> blk;
            for (UINT32 p = 0; p < blocks_[k].keep.size(); ++p) blk.push_back(gid(k, p));
            std::vector<UINT32> U = prev;
            U.insert(U.end(), blk.begin(), blk.end());
            std::sort(U.begin(), U.end());
            U.erase(std::unique(U.begin(), U.end()), U.end());
            for (UINT32 s : prev) st.pos_prev.push_back((UINT32)position(U, s));
            for (UINT32 s : blk) st.pos_blk.push_back((UINT32)position(U, s));
            // stations that stay: the run's first junction row and block k's junction row towards k + 1
            std::vector<UINT32> stay;
            for (UINT32 p : blocks_[g.a].c_prev) stay.push_back(gid(g.a, p));
            if (k + 1 < blockCount_)
                for (UINT32 p : blocks_[k].c_next) stay.push_back(gid(k, p));
            std::sort(stay.begin(), stay.end());
            stay.erase(std::unique(stay.begin(), stay.end()), stay.end());
            for (UINT32 s : stay) st.keep.push_back((UINT32)position(U, s));
            // constraints of the stations that leave inside the run: where the forward chain adds them (first appearance)
            for (UINT32 kk : (k == g.a + 1 ? std::vector<UINT32>{g.a, k} : std::vector<UINT32>{k})) {
                const constraint_list& src = blocks_[kk].ccon_fwd;
                for (size_t i = 0; i < src.stn.size(); ++i) {
                    const UINT32 s = gid(kk, src.stn[i]);
                    if (position(g.stations, s) >= 0) continue;
                    st.con.stn.push_back((UINT32)position(U, s));
                    st.con.w9.insert(st.con.w9.end(), src.w9.begin() + 9 * i, src.w9.begin() + 9 * i + 9);
                }
            }
            st.n_stn = (UINT32)U.size();
            st.dev_block = 2 * blockCount_ + k;
            NewBlock(st.dev_block, st.n_stn, 0, k, "PrepareAdjustment(): run merge");
            max_keep = std::max(max_keep, stay.size());
            prev = stay;
            g.steps.push_back(std::move(st));
        }
        if (g.a != g.b && prev != g.stations) {       // (the last merge must leave exactly the run's end stations)
            FreeTwoLevel();
            return;
        }
        if (g.steps.size() > 1)
            for (dnagpu_matrix*& m : g.M) NewMatrix((UINT32)max_keep * 3, &m, g.a, "PrepareAdjustment(): run merge");
    }
    two_level_ok_ = true;
}

// level 1: the own run condensed to its end stations
void dna_adjust::ReduceOwnRun() { ReduceRun(0, DistRank()); }

void dna_adjust::ReduceRun(int c, int run) {
    segment_t& g = segs_[run];
    if (g.a == g.b) {
        Check(dnagpu_matrix_copy(ctx_, c, g.S, blocks_[g.a].red), g.a, "Solve()");
        Check(dnagpu_chain_sync(ctx_, c), g.a, "Solve()");
        return;
    }
    const dnagpu_matrix* prev = blocks_[g.a].red;
    for (size_t i = 0; i < g.steps.size(); ++i) {
        const seg_step_t& st = g.steps[i];
        const UINT32 k = g.a + 1 + (UINT32)i;
        currentBlock_ = k;
        dnagpu_matrix* Wm = work_[c];
        Check(dnagpu_matrix_reset(ctx_, c, Wm, 3 * st.n_stn), k, "UpdateNormals()");
        Check(dnagpu_junction_scatter(ctx_, c, Wm, st.pos_prev.data(), st.pos_prev.size(), prev), k, "UpdateNormals()");
        Check(dnagpu_junction_scatter(ctx_, c, Wm, st.pos_blk.data(), st.pos_blk.size(), blocks_[k].red), k, "UpdateNormals()");
        Check(dnagpu_block_add_rhs(ctx_, c, st.dev_block, st.pos_prev.data(), st.pos_prev.size(), prev, 1), k, "Solve()");
        Check(dnagpu_block_add_rhs(ctx_, c, st.dev_block, st.pos_blk.data(), st.pos_blk.size(), blocks_[k].red, 0), k, "Solve()");
        AddConstraints(c, Wm, st.con, +1, k);
        dnagpu_matrix* out = (i + 1 == g.steps.size()) ? g.S : g.M[i & 1];
        Check(dnagpu_block_reduce(ctx_, c, st.dev_block, Wm, st.keep.data(), st.keep.size(), out, nullptr), k, "Solve()");
        prev = out;
        const double nk = 3.0 * (double)st.keep.size(), ni = 3.0 * (double)st.n_stn - nk;
        std::lock_guard<std::mutex> lk(corr_mutex_);
        CountFlops(ni * ni * ni / 3.0 + ni * ni * nk + ni * nk * nk, 0);
    }
}

void dna_adjust::ExchangeRuns() {
    const double t0 = wall_ms();
    struct part_t { double *F, *v; UINT32 np; };
    std::vector<part_t> parts(DistWorld());
    AgreeOnPhase("exchange of the run systems (preparation)", [&] {        // (see ExchangeCondensed)
        for (int r = 0; r < DistWorld(); ++r) {
            segment_t& g = segs_[r];
            Check(dnagpu_matrix_resize(ctx_, g.S, (UINT32)g.stations.size() * 3), g.a, "exchange");
            dnagpu_matrix_device_pointers(g.S, &parts[r].F, &parts[r].v, &parts[r].np);
        }
    });
    comm_->group_begin();
    for (int r = 0; r < DistWorld(); ++r) {
        comm_->broadcast(parts[r].F, (size_t)parts[r].np * parts[r].np, r);
        comm_->broadcast(parts[r].v, parts[r].np, r);
    }
    comm_->group_end();
    comm_->wait();
    exchange_ms_ += wall_ms() - t0;
}

// level 2: the two chains over the runs (every rank, identical arithmetic): jfwd at the last block of every run but the last,
// jrev at the block before every run but the first
void dna_adjust::ScanRuns() {
    const int W = (int)segs_.size();
    const bool two = NumChains() > 1;
    auto load = [&](int c, segment_t& g) {
        dnagpu_matrix* Wm = work_[c];
        Check(dnagpu_block_gather_stations(ctx_, c, g.dev_block, g.dstA.data(), g.a, g.srcA.data(), g.dstA.size()), g.a, "UpdateNormals()");
        Check(dnagpu_block_gather_stations(ctx_, c, g.dev_block, g.dstB.data(), g.b, g.srcB.data(), g.dstB.size()), g.b, "UpdateNormals()");
        Check(dnagpu_matrix_copy(ctx_, c, Wm, g.S), g.a, "UpdateNormals()");
        std::vector<UINT32> all(g.stations.size());
        std::iota(all.begin(), all.end(), 0u);
        Check(dnagpu_block_add_rhs(ctx_, c, g.dev_block, all.data(), all.size(), g.S, 1), g.a, "Solve()");
        return Wm;
    };
    auto carry = [&](int c, segment_t& g, dnagpu_matrix* Wm, const std::vector<UINT32>& out, dnagpu_matrix* jm, UINT32 k) {
        Check(dnagpu_schur_carry(ctx_, c, g.dev_block, Wm, out.data(), out.size(), jm), k, "Solve()");
        const double n = 3.0 * (double)g.stations.size(), nj = 3.0 * (double)out.size(), ni = n - nj;
        std::lock_guard<std::mutex> lk(corr_mutex_);
        CountFlops(ni * ni * ni / 3.0 + ni * ni * nj + ni * nj * nj + (dnagpu_info_carry(ctx_) ? 0.0 : nj * nj * nj), 0);
    };
    OnEveryChain([&](int c) {
        if (c == 0)
            for (int r = 0; r + 1 < W && !IsCancelled() && !chain_failed_; ++r) {
                segment_t& g = segs_[r];
                dnagpu_matrix* Wm = load(c, g);
                AddConstraints(c, Wm, g.con_fwd, +1, g.a);
                if (r > 0) {
                    Check(dnagpu_junction_scatter(ctx_, c, Wm, g.posL.data(), g.posL.size(), blocks_[g.a - 1].jfwd), g.a, "CarryStnEstimatesandVariancesForward()");
                    Check(dnagpu_junction_rhs(ctx_, c, g.dev_block, g.posL.data(), g.posL.size(), blocks_[g.a - 1].jfwd), g.a, "Solve()");
                }
                carry(c, g, Wm, g.posR, blocks_[g.b].jfwd, g.b);
            }
        if (c == 1 || !two)
            for (int r = W - 1; r >= 1 && !IsCancelled() && !chain_failed_; --r) {
                segment_t& g = segs_[r];
                dnagpu_matrix* Wm = load(c, g);
                if (r + 1 < W) Check(dnagpu_junction_scatter(ctx_, c, Wm, g.posR.data(), g.posR.size(), blocks_[g.b].jrev), g.b, "CarryStnEstimatesandVariancesReverse()");
                AddConstraints(c, Wm, g.con_rev, +1, g.b);
                if (r + 1 < W) Check(dnagpu_junction_rhs(ctx_, c, g.dev_block, g.posR.data(), g.posR.size(), blocks_[g.b].jrev), g.b, "Solve()");
                carry(c, g, Wm, g.posL, blocks_[g.a - 1].jrev, g.a);
            }
    });
}

// level 3: both chains over the own blocks, from the boundary values of level 2
void dna_adjust::OwnRunChains() {
    const segment_t& g = segs_[DistRank()];
    const bool two = NumChains() > 1;
    OnEveryChain([&](int c) {
        if (c == 0)
            for (UINT32 k = g.a; k < g.b && !IsCancelled() && !chain_failed_; ++k) CondensedForwardBlock(c, k);
        if (c == 1 || !two)
            for (UINT32 k = g.b; k > g.a && !IsCancelled() && !chain_failed_; --k) CondensedReverseBlock(c, k);
    });
}

void dna_adjust::DistributedCondensedIteration() {
    std::vector<UINT32> mine;
    for (UINT32 k = 0; k < blockCount_; ++k)
        if (OwnsBlock(k)) mine.push_back(k);
    AgreeOnPhase("condensing the blocks", [&] { CondenseBlocks(mine); });
    if (cancel_agreed_) return;
    if (two_level_ok_) {
        auto timed = [&](const std::function<void()>& body) {
            const double t0 = wall_ms();
            body();
            Check(dnagpu_sync(ctx_), 0, "AdjustNetwork()");
            chain_ms_ += wall_ms() - t0;
        };
        AgreeOnPhase("junction chains (own run)", [&] { timed([&] { ReduceOwnRun(); }); });
        ExchangeRuns();
        AgreeOnPhase("junction chains", [&] {
            timed([&] {
                ScanRuns();
                OwnRunChains();
            });
        });
        if (cancel_agreed_) return;
        AgreeOnPhase("rigorous block solutions", [&] { RigorousBlocks(mine); });
        return;
    }
    ExchangeCondensed();
    const double t0 = wall_ms();
    AgreeOnPhase("junction chains", [&] { CondensedChains(); });
    chain_ms_ += wall_ms() - t0;
    if (cancel_agreed_) return;
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
    if (cancel_agreed_) return;
    // the junction payloads of the combination solves: jfwd[k-1] from the forward rank, jrev[k] from the reverse rank
    const double t0 = wall_ms();
    std::vector<junction_msg> msgs;
    for (UINT32 k = 0; k < blockCount_; ++k) {
        if (!CombineRequired(k)) continue;
        const int o = BlockOwner(k);
        if (fwd_rank != o) msgs.push_back({0, k - 1, fwd_rank, o});
        if (rev_rank != o) msgs.push_back({1, k, rev_rank, o});
    }
    // Every junction that travels here left a step that carries (PhasedForwardBlock / PhasedReverseBlock): with a.schur_carry it was made by
    // elimination and is in dnagpu_schur_carry's form -- by default the information form: matrix, linearisation point AND reduced right-hand
    // side --, otherwise gathered from the block inverse and inverted (estimates form).  Both sides know which: one rule, same settings.
    const int form = (SchurCarry() && dnagpu_info_carry(ctx_)) ? 1 : 0;
    struct xfer_t { double *F, *v, *r; UINT32 np; int peer; bool send; };
    std::vector<xfer_t> xfers;
    AgreeOnPhase("exchange of the junction matrices (preparation)", [&] {       // (see ExchangeCondensed)
        for (const junction_msg& g : msgs) {
            if (g.src != me && g.dst != me) continue;
            dnagpu_matrix* jm = g.kind == 0 ? blocks_[g.block].jfwd : blocks_[g.block].jrev;
            if (!jm) continue;
            if (g.dst == me) Check(dnagpu_matrix_resize(ctx_, jm, JunctionUnknowns(g.block)), g.block, "exchange");
            xfer_t x{nullptr, nullptr, nullptr, 0, g.src == me ? g.dst : g.src, g.src == me};
            int has = 0;
            Check(dnagpu_junction_device_pointers(ctx_, jm, g.dst == me ? form : -1, &x.F, &x.v, &x.r, &x.np, &has), g.block, "exchange");
            if (has != form) SignalExceptionAdjustment("AdjustPhased(): a junction matrix is not in the form the exchange expects.", g.block);
            xfers.push_back(x);
        }
    });
    comm_->group_begin();
    for (const xfer_t& x : xfers) {
        if (x.send) {
            comm_->send(x.F, (size_t)x.np * x.np, x.peer);
            comm_->send(x.v, x.np, x.peer);
            if (x.r) comm_->send(x.r, x.np, x.peer);
        } else {
            comm_->recv(x.F, (size_t)x.np * x.np, x.peer);
            comm_->recv(x.v, x.np, x.peer);
            if (x.r) comm_->recv(x.r, x.np, x.peer);
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
    cancel_agreed_ = false;
    for (UINT32 i = 0; i < projectSettings_.a.max_iterations; ++i) {
        // CancelAdjustment() may have reached one rank only (its own process, its own signal): the ranks agree before anyone leaves
        AgreeOnPhase("start of the iteration", [] {});
        if (cancel_agreed_) break;
        const double it_t0 = wall_ms();
        PhasedBeginIteration();
        if (CondensedSchedule())
            DistributedCondensedIteration();
        else
            DistributedReferenceIteration();
        if (cancel_agreed_) break;
        Check(dnagpu_sync(ctx_), 0, "AdjustNetwork()");
        SyncCoordinates();
        if (cancel_agreed_) break;
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
    double* dev = nullptr;
    AgreeOnPhase("statistics (preparation)", [&] {          // (see ExchangeCondensed)
        dev = ExchangeBuffer(host.size());
        Check(dnagpu_copy(ctx_, dev, host.data(), host.size() * sizeof(double)), 0, "exchange");
    });
    comm_->all_reduce_sum(dev, host.size());
    comm_->wait();
    AgreeOnPhase("statistics (collection)", [&] { Check(dnagpu_copy(ctx_, host.data(), dev, host.size() * sizeof(double)), 0, "exchange"); });
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
    // the two ranks involved prepare (the device buffer, the owner's copy of the matrix out of HBM or the staging area); every rank
    // takes part in the agreement, so that a failure there leaves nobody in ncclSend / ncclRecv
    double* dev = nullptr;
    std::vector<double> tail(rows + 1, 0.0);
    AgreeOnPhase("collecting the block results (preparation)", [&] {
        if (me != 0 && me != o) return;
        dev = ExchangeBuffer(cnt + rows + 1);
        if (me == o) {
            GetBlockRigorousVariancesPacked(b, packed);
            tail[0] = blocks_[b].prec_adj_msrs.size() == rows ? 1.0 : 0.0;
            if (tail[0] > 0.5) std::copy(blocks_[b].prec_adj_msrs.begin(), blocks_[b].prec_adj_msrs.end(), tail.begin() + 1);
            Check(dnagpu_copy(ctx_, dev, packed.data(), cnt * sizeof(double)), b, "exchange");
            Check(dnagpu_copy(ctx_, dev + cnt, tail.data(), tail.size() * sizeof(double)), b, "exchange");
        }
    });
    if (me != 0 && me != o) return;
    if (me == o) {
        comm_->send(dev, cnt + rows + 1, 0);
        comm_->wait();
        packed.clear();
    } else {
        comm_->recv(dev, cnt + rows + 1, o);
        comm_->wait();
        packed.resize(cnt);
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
            a.shares_device_ = !distinct;          // several contexts on one GPU: no fused (mutually waiting) launches
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
