// Tile-DAG executor (tile_dag.h): builder (host), kernel, CPU execution for the tests.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include "tile_dag.h"
#include "gemm_tile_dma.h"
#include "gemm_tile_reg.h"
#include "leaf_body.h"

namespace dnagpu {

// ------------------------------------------------------------------------------------------------------------------------------
// Builder
// ------------------------------------------------------------------------------------------------------------------------------
static constexpr uintptr_t FAKE_SHIFT = 40;      // symbolic buffers 1 TiB apart

DagBuilder::DagBuilder(int nbuf, const int* ld, long small_tiles) : nbuf_(nbuf), small_tiles_(small_tiles) {
    for (int b = 0; b < DAG_MAX_BUFS; ++b) ld_[b] = b < nbuf ? ld[b] : 128;
}

double* DagBuilder::base(int b) const { return reinterpret_cast<double*>((uintptr_t)(b + 1) << FAKE_SHIFT); }

bool DagBuilder::decode(const void* p, int& buf, uint32_t& off, int& rt, int& ct) const {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    buf = (int)(a >> FAKE_SHIFT) - 1;
    if (buf < 0 || buf >= nbuf_) return false;
    const uint64_t e = (a & (((uintptr_t)1 << FAKE_SHIFT) - 1)) / sizeof(double);
    if (e >> 32) return false;
    off = (uint32_t)e;
    const uint64_t col = e / (uint64_t)ld_[buf], row = e % (uint64_t)ld_[buf];
    if ((col & 127) || (row & 127)) return false;
    rt = (int)(row / 128);
    ct = (int)(col / 128);
    return true;
}

void DagBuilder::add_gemm(const GemmArgs& a, int akc, int bkc) {
    Op o{};
    int ab, bb, cb;
    if (!decode(a.A, ab, o.a_off, o.a_rt, o.a_ct) || !decode(a.B, bb, o.b_off, o.b_rt, o.b_ct) || !decode(a.C, cb, o.c_off, o.c_rt, o.c_ct) ||
        a.lda != ld_[ab] || a.ldb != ld_[bb] || a.ldc != ld_[cb] || (a.K & 127) || (akc && !bkc)) {
        fprintf(stderr, "dnagpu: tile DAG recording: unsupported product\n");
        abort();
    }
    o.type = akc ? DAG_GEMM_TN : (bkc ? DAG_GEMM_NN : DAG_GEMM_NT);
    const long total = a.lower ? (long)a.mt * (a.mt + 1) / 2 : (long)a.mt * a.nt;
    const int sub = total < small_tiles_ ? 2 : 1;               // 64 x 64 tasks: two per 128-tile and direction
    if (sub == 2) o.type |= DAG_TILE64;
    o.bufs = (uint8_t)(ab | (bb << 2) | (cb << 4));
    o.flags = (uint8_t)((a.alpha < 0.0 ? DAG_ALPHA_NEG : 0) | (a.beta != 0.0 ? DAG_BETA_ONE : 0) | (a.mirror ? DAG_MIRROR : 0) |
                        ((a.kmode == KM_GE_J || a.kmode == KM_GE_I) ? DAG_DOWN : 0));
    if (std::fabs(a.alpha) != 1.0 || (a.beta != 0.0 && a.beta != 1.0)) {
        fprintf(stderr, "dnagpu: tile DAG recording: alpha / beta outside {+-1} / {0, 1}\n");
        abort();
    }
    o.mt = a.mt; o.nt = a.nt; o.kt = a.K / 128; o.kmode = a.kmode; o.lower = a.lower;
    o.id_base = nids_;
    nids_ += (uint32_t)(a.mt * sub) * (uint32_t)(a.nt * sub);
    ops_.push_back(o);
}

void DagBuilder::add_leaf(const double* A_tile, double* X_tile, int diag_tile) {
    Op o{};
    int ab, cb;
    if (!decode(A_tile, ab, o.a_off, o.a_rt, o.a_ct) || !decode(X_tile, cb, o.c_off, o.c_rt, o.c_ct)) {
        fprintf(stderr, "dnagpu: tile DAG recording: unsupported leaf\n");
        abort();
    }
    o.type = DAG_LEAF;
    o.bufs = (uint8_t)(ab | (cb << 4));
    o.mt = o.nt = 1;
    o.kt = diag_tile;
    o.id_base = nids_++;
    ops_.push_back(o);
}

namespace {

// A 128 x 128 tile of a buffer: who wrote it last and who has read it since, as completion flags.  The tasks of ONE product write
// disjoint parts of a tile (a 128-tile task the whole of it, 64-tile tasks a quadrant each) and never read what a sibling writes,
// so they form one generation: `writers` are the siblings so far, prev_* the generation before, which every sibling has to respect.
struct TileState {
    int32_t op = -1;                     // the product the current generation of writers belongs to
    std::vector<uint32_t> writers, readers, prev_writers, prev_readers;
};

struct TileMap {
    int rows = 1;
    std::vector<TileState> t;
    TileState& at(int rt, int ct) {
        const size_t i = (size_t)ct * rows + rt;
        if (i >= t.size()) t.resize(std::max(i + 1, t.size() * 2));
        return t[i];
    }
};

// model of a task's duration (microseconds): what the critical-path priorities and the diagnostic simulation are computed with
// (measured inside the launch, DNAGPU_DAG_TRACE at n = 20 000: 128-tile tasks 30 us at one k-tile and 20.4 - 22.8 us per k-tile beyond,
//  64-tile tasks 9 - 10 us per k-tile, a leaf of four waves 58 - 60 us)
inline float task_us(const DagTask& t) {
    if (t.type == DAG_LEAF) return 60.0f;
    const float nk = (float)(t.ke > t.kb ? t.ke - t.kb : 0);
    return (t.type & DAG_TILE64) ? 10.0f * nk : 8.0f + 22.0f * nk;
}

// recorded numbers (sorted, unique) -> arithmetic runs
void compress(const std::vector<uint32_t>& p, std::vector<DagRun>& out) {
    size_t i = 0;
    while (i < p.size()) {
        DagRun d{p[i], 1, 1};
        if (i + 1 < p.size()) {
            const uint32_t stride = p[i + 1] - p[i];
            if (stride <= 0xffffu) {
                size_t j = i + 1;
                while (j < p.size() && p[j] - p[j - 1] == stride && d.count < 64) {       // (one task per lane of the wave that walks the run)
                    ++d.count;
                    ++j;
                }
                d.stride = (uint16_t)stride;
            }
        }
        out.push_back(d);
        i += d.count;
    }
}

}  // namespace

std::shared_ptr<DagGraph> DagBuilder::finish(int workers) {
    auto g = std::make_shared<DagGraph>();
    TileMap tiles[DAG_MAX_BUFS];
    for (int b = 0; b < DAG_MAX_BUFS; ++b) tiles[b].rows = std::max(1, ld_[b] / 128);
    std::vector<DagTask> T;                                 // in recorded order
    std::vector<uint32_t> pred_off(1, 0), pred_idx;      // predecessors as task numbers (program order), for the scheduling
    std::vector<int32_t> id2rec(nids_, -1);                 // recorded number -> index in T
    std::vector<uint32_t> preds;
    std::vector<TileState*> ins;         // tiles read (operands)
    std::vector<TileState*> outs;        // tiles written
    int32_t op_id = -1;
    for (const Op& o : ops_) {
        ++op_id;
        const int ab = o.bufs & 3, bb = (o.bufs >> 2) & 3, cb = (o.bufs >> 4) & 3;
        const bool leaf = o.type == DAG_LEAF;
        const int sub = (!leaf && (o.type & DAG_TILE64)) ? 2 : 1;
        const int variant = o.type & 3;
        const int mts = o.mt * sub, nts = o.nt * sub;
        for (int it = 0; it < mts; ++it) {
            const int jmax = (!leaf && o.lower) ? it : nts - 1;
            for (int jt = 0; jt <= jmax; ++jt) {
                const int bi = it / sub, bj = jt / sub;          // the 128-tile the task's tile lies in
                DagTask t{};
                t.a_off = o.a_off; t.b_off = o.b_off; t.c_off = o.c_off;
                t.it = (uint16_t)it; t.jt = (uint16_t)jt;
                t.type = o.type; t.bufs = o.bufs; t.flags = o.flags;
                t.id = o.id_base + (uint32_t)it * (uint32_t)nts + (uint32_t)jt;
                int kb = 0, ke = 0;
                if (leaf) {
                    t.kb = t.ke = (uint16_t)o.kt;
                    g->n_leaves++;
                } else {
                    ke = o.kt;
                    switch (o.kmode) {
                        case KM_LE_J: ke = bj + 1; break;
                        case KM_GE_J: kb = bj; break;
                        case KM_LE_I: ke = bi + 1; break;
                        case KM_GE_I: kb = bi; break;
                        default: break;
                    }
                    if (ke > o.kt) ke = o.kt;
                    if (ke < kb) ke = kb;
                    t.kb = (uint16_t)kb; t.ke = (uint16_t)ke;
                    if (!((o.flags & DAG_MIRROR) && it != jt)) t.flags &= (uint8_t)~DAG_MIRROR;
                    g->flops += 2.0 * (128.0 / sub) * (128.0 / sub) * 128.0 * (double)(ke - kb);
                }
                // the tiles the task reads and writes (collected twice: TileMap::at may grow a map, which moves its tiles)
                TileState* own_c = nullptr;      // beta = 1: its own part of C, read before it is written
                for (int pass = 0; pass < 2; ++pass) {
                    ins.clear();
                    outs.clear();
                    if (leaf) {
                        ins.push_back(&tiles[ab].at(o.a_rt, o.a_ct));
                        outs.push_back(&tiles[cb].at(o.c_rt, o.c_ct));
                        continue;
                    }
                    const bool akc = variant == DAG_GEMM_TN, bkc = variant != DAG_GEMM_NT;
                    for (int k = kb; k < ke; ++k) {
                        ins.push_back(akc ? &tiles[ab].at(o.a_rt + k, o.a_ct + bi) : &tiles[ab].at(o.a_rt + bi, o.a_ct + k));
                        ins.push_back(bkc ? &tiles[bb].at(o.b_rt + k, o.b_ct + bj) : &tiles[bb].at(o.b_rt + bj, o.b_ct + k));
                    }
                    own_c = (o.flags & DAG_BETA_ONE) ? &tiles[cb].at(o.c_rt + bi, o.c_ct + bj) : nullptr;
                    outs.push_back(&tiles[cb].at(o.c_rt + bi, o.c_ct + bj));
                    if (t.flags & DAG_MIRROR) outs.push_back(&tiles[cb].at(o.c_rt + bj, o.c_ct + bi));
                }
                preds.clear();
                for (TileState* in : ins) {                                                          // read after write
                    const std::vector<uint32_t>& w = in->op == op_id ? in->prev_writers : in->writers;
                    preds.insert(preds.end(), w.begin(), w.end());
                }
                if (own_c) {
                    const std::vector<uint32_t>& w = own_c->op == op_id ? own_c->prev_writers : own_c->writers;
                    preds.insert(preds.end(), w.begin(), w.end());
                }
                for (TileState* w : outs) {
                    const bool sibling = w->op == op_id;
                    const std::vector<uint32_t>& pw = sibling ? w->prev_writers : w->writers;        // write after write
                    const std::vector<uint32_t>& pr = sibling ? w->prev_readers : w->readers;        // write after read
                    preds.insert(preds.end(), pw.begin(), pw.end());
                    if (!getenv("DNAGPU_DAG_TEST_DROP_WAR")) preds.insert(preds.end(), pr.begin(), pr.end());   // (test hook: tests/test_tile_dag.py shows the self-test notices)
                }
                std::sort(preds.begin(), preds.end());
                preds.erase(std::unique(preds.begin(), preds.end()), preds.end());
                preds.erase(std::remove(preds.begin(), preds.end(), t.id), preds.end());
                for (TileState* in : ins)
                    if (in->op != op_id && (in->readers.empty() || in->readers.back() != t.id)) in->readers.push_back(t.id);
                for (TileState* w : outs) {
                    if (w->op != op_id) {
                        w->prev_writers.swap(w->writers);
                        w->prev_readers.swap(w->readers);
                        w->writers.clear();
                        w->readers.clear();
                        w->op = op_id;
                    }
                    if (w->writers.empty() || w->writers.back() != t.id) w->writers.push_back(t.id);
                }
                id2rec[t.id] = (int32_t)T.size();
                for (uint32_t f : preds) pred_idx.push_back((uint32_t)id2rec[f]);
                pred_off.push_back((uint32_t)pred_idx.size());
                T.push_back(t);
            }
        }
        if (!leaf) g->n_products++;
    }
    g->nids = nids_;
    const size_t N = T.size();

    // ---- critical-path lengths (the priorities) with the duration model ----
    std::vector<float> dur(N), bl(N);
    for (size_t i = 0; i < N; ++i) {
        bl[i] = dur[i] = task_us(T[i]);
        g->sim_work_us += dur[i];
    }
    for (size_t i = N; i-- > 0;)
        for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) {
            const uint32_t p = pred_idx[q];
            bl[p] = std::max(bl[p], dur[p] + bl[i]);
        }
    for (size_t i = 0; i < N; ++i) g->critical_path_us = std::max(g->critical_path_us, (double)bl[i]);

    // ---- successors (the device's view of the dependencies) ----
    std::vector<uint32_t> succ_off(N + 1, 0), succ_idx(pred_idx.size()), indeg(N);
    for (size_t i = 0; i < N; ++i) {
        indeg[i] = pred_off[i + 1] - pred_off[i];
        for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) succ_off[pred_idx[q] + 1]++;
    }
    for (size_t i = 0; i < N; ++i) succ_off[i + 1] += succ_off[i];
    {
        std::vector<uint32_t> fill(succ_off.begin(), succ_off.end() - 1);
        for (size_t i = 0; i < N; ++i)
            for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) succ_idx[fill[pred_idx[q]]++] = (uint32_t)i;      // (ascending i: recorded order)
    }

    // ---- ready queues: sixteen classes of remaining path, the longest first ----
    const double cp = std::max(1.0, g->critical_path_us);
    for (size_t i = 0; i < N; ++i) {
        int q = (int)((1.0 - (double)bl[i] / cp) * DAG_QUEUES);
        T[i].queue = (uint8_t)std::min(DAG_QUEUES - 1, std::max(0, q));
    }
    std::vector<uint32_t> order(N);
    for (size_t i = 0; i < N; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (T[x].queue != T[y].queue) return T[x].queue < T[y].queue;
        return bl[x] > bl[y];
    });
    std::vector<uint32_t> rec2task(N);
    g->tasks.resize(N);
    g->id2task.assign(nids_, 0xffffffffu);
    for (size_t k = 0; k < N; ++k) {
        g->tasks[k] = T[order[k]];
        rec2task[order[k]] = (uint32_t)k;
        g->id2task[T[order[k]].id] = (uint32_t)k;
    }
    for (int q = 0; q <= DAG_QUEUES; ++q) g->slot_base[q] = 0;
    for (size_t k = 0; k < N; ++k) g->slot_base[g->tasks[k].queue + 1]++;
    for (int q = 0; q < DAG_QUEUES; ++q) g->slot_base[q + 1] += g->slot_base[q];
    std::vector<uint32_t> ids;
    for (size_t k = 0; k < N; ++k) {
        const uint32_t i = order[k];
        ids.clear();
        for (uint32_t q = succ_off[i]; q < succ_off[i + 1]; ++q) ids.push_back(T[succ_idx[q]].id);
        std::sort(ids.begin(), ids.end());
        g->tasks[k].succ0 = (uint32_t)g->succ.size();
        compress(ids, g->succ);
        g->tasks[k].nsucc = (uint32_t)g->succ.size() - g->tasks[k].succ0;
    }

    // ---- the state image a launch starts from ----
    g->state_pending = DAG_STATE_FIXED;
    g->state_slots = g->state_pending + nids_;
    g->state_mail = g->state_slots + (uint32_t)N;
    g->state_init.assign((size_t)g->state_mail + N, 0u);
    for (int q = 0; q <= DAG_QUEUES; ++q) g->state_init[2 * DAG_QUEUES + q] = g->slot_base[q];
    for (size_t i = 0; i < N; ++i) g->state_init[g->state_pending + T[i].id] = indeg[i];
    for (size_t k = 0; k < N; ++k)
        if (!indeg[order[k]]) {
            const int q = g->tasks[k].queue;
            uint32_t& tail = g->state_init[DAG_QUEUES + q];
            g->state_init[g->state_slots + g->slot_base[q] + tail] = (uint32_t)k + 1;
            ++tail;
            ++g->state_init[DAG_STATE_BALANCE];              // tasks queued and not yet promised to a workgroup
        }

    // ---- diagnostic: list scheduling on `workers` workgroups with the duration model ----
    if (N > 1) {
        struct Ready {
            float bl;
            uint32_t i;
            bool operator<(const Ready& o) const { return bl != o.bl ? bl < o.bl : i > o.i; }
        };
        std::priority_queue<Ready> ready;
        typedef std::pair<double, uint32_t> Fin;
        std::priority_queue<Fin, std::vector<Fin>, std::greater<Fin>> running;
        std::vector<uint32_t> left(indeg);
        for (size_t i = 0; i < N; ++i)
            if (!left[i]) ready.push({bl[i], (uint32_t)i});
        double now = 0.0;
        int free_w = std::max(1, workers);
        size_t started = 0;
        while (started < N) {
            while (free_w > 0 && !ready.empty()) {
                const uint32_t i = ready.top().i;
                ready.pop();
                ++started;
                running.push({now + dur[i], i});
                --free_w;
            }
            if (running.empty()) break;
            now = running.top().first;
            while (!running.empty() && running.top().first <= now) {
                const uint32_t i = running.top().second;
                running.pop();
                ++free_w;
                for (uint32_t q = succ_off[i]; q < succ_off[i + 1]; ++q)
                    if (--left[succ_idx[q]] == 0) ready.push({bl[succ_idx[q]], succ_idx[q]});
            }
        }
        while (!running.empty()) {
            now = running.top().first;
            running.pop();
        }
        g->sim_makespan_us = now;
    }
    return g;
}

DagGraph::~DagGraph() {
    if (d_tasks || d_succ || d_id2task || d_state_init) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (device >= 0 && device != cur) (void)hipSetDevice(device);
        for (void* p : {(void*)d_tasks, (void*)d_succ, (void*)d_id2task, (void*)d_state_init})
            if (p) (void)hipFree(p);
        if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    }
}

hipError_t dag_upload(DagGraph& g) {
    if (g.d_tasks) return hipSuccess;
    hipError_t e = hipGetDevice(&g.device);
    if (e != hipSuccess) return e;
    auto up = [&](auto*& dev, const auto& host) {
        typedef typename std::remove_reference<decltype(host)>::type::value_type V;
        hipError_t r = hipMalloc(&dev, std::max<size_t>(1, host.size()) * sizeof(V));
        if (r == hipSuccess && !host.empty()) r = hipMemcpy(dev, host.data(), host.size() * sizeof(V), hipMemcpyHostToDevice);
        return r;
    };
    if ((e = up(g.d_tasks, g.tasks)) != hipSuccess) return e;
    if ((e = up(g.d_succ, g.succ)) != hipSuccess) return e;
    if ((e = up(g.d_id2task, g.id2task)) != hipSuccess) return e;
    return up(g.d_state_init, g.state_init);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Kernel
// ------------------------------------------------------------------------------------------------------------------------------
// LDS: the tile product's two operand buffers (distinct objects, gemm_tile_dma.h) double as the home of a leaf's 36 blocks:
// lds0 = LT (256 doubles) + blocks 0..15, lds1 = blocks 16..32 (16 doubles longer than the product needs), lds2 = blocks 33..35:
// 80 384 bytes, two workgroups per CU like the per-product kernels.
__global__ __launch_bounds__(256, 2) void tile_dag_kernel(DagLaunch L) {
    using G = Geo<128, 4>;
    __shared__ __attribute__((aligned(16))) double lds0[2 * G::OPBUF];
    __shared__ __attribute__((aligned(16))) double lds1[2 * G::OPBUF + 16];
    __shared__ __attribute__((aligned(16))) double lds2[3 * leaf::BS];
    __shared__ uint32_t s_task;
    static_assert(2 * G::OPBUF == 256 + 16 * leaf::BS && 2 * G::OPBUF + 16 == 17 * leaf::BS, "leaf blocks in the operand buffers");
    const int tid = threadIdx.x;
    unsigned long long t_start = 0;
    if (L.trace && tid == 0) t_start = wall_clock64();

    // ---- take ONE ready task (wave 0) ----
    // `balance` = tasks put into the queues minus workgroups arrived, changed by ONE atomic per arrival and per ready task, which
    // puts all of them into one order.  A workgroup that finds it positive has a queue entry to itself (it may have to look twice
    // until the entry's writer is done); one that finds it <= 0 is the w-th workgroup to wait and the w-th task that becomes ready
    // while somebody waits is written straight into its own mailbox word -- nobody ever polls a word that others poll too.
    uint32_t* const heads = L.state;
    uint32_t* const tails = L.state + DAG_QUEUES;
    if (tid < 64) {
        uint32_t task = 0xffffffffu;
        int old = 0;
        uint32_t w = 0;
        if (tid == 0) {
            old = __hip_atomic_fetch_sub((int*)(L.state + DAG_STATE_BALANCE), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old <= 0) w = __hip_atomic_fetch_add(L.state + DAG_STATE_WAITERS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        long spins = 0;
        if (old > 0) {
            for (;;) {
                uint32_t h = 0, tl = 0;
                if (tid < DAG_QUEUES) {
                    h = __hip_atomic_load(heads + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tl = __hip_atomic_load(tails + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned long long avail = __ballot(tid < DAG_QUEUES && (int32_t)(tl - h) > 0);
                if (avail) {
                    const int q = __builtin_ctzll(avail);                  // the most urgent queue with an entry
                    const uint32_t hq = (uint32_t)__builtin_amdgcn_readlane((int)h, q);
                    int won = 0;
                    if (tid == 0) won = atomicCAS(heads + q, hq, hq + 1) == hq;
                    if (__builtin_amdgcn_readfirstlane(won)) {
                        const uint32_t* slot = L.state + L.state_slots + L.state[2 * DAG_QUEUES + q] + hq;
                        uint32_t v;
                        while ((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1);
                        task = v - 1;
                        break;
                    }
                    continue;                                               // another workgroup was faster: look again
                }
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1L << 22)) {
                    if (tid == 0) atomicMin(L.info, INFO_BARRIER_TIMEOUT);
                    break;
                }
            }
        } else {
            const uint32_t* mail = L.state + L.state_mail + w;
            uint32_t v;
            while ((v = __hip_atomic_load(mail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
                // seconds without a task: one never finished (a bug, or a workgroup of this launch that died).  The first workgroup to give
                // up says so in `info`, the others see that and give up too: the launch ends with a void result instead of hanging
                if (spins > (1L << 22) || ((spins & 1023) == 0 && __hip_atomic_load(L.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == INFO_BARRIER_TIMEOUT)) {
                    if (tid == 0) atomicMin(L.info, INFO_BARRIER_TIMEOUT);
                    break;
                }
            }
            if (v) task = v - 1;
        }
        if (tid == 0) s_task = task;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // this CU's L1 forgets what it held of the predecessors' tiles
    }
    __syncthreads();
    const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_task);
    if (ticket >= L.ntasks) return;
    const DagTask* tp = L.tasks + ticket;
    const uint32_t a_off = tp->a_off, b_off = tp->b_off, c_off = tp->c_off;
    const int it = tp->it, jt = tp->jt, kb = tp->kb, ke = tp->ke;
    const int type = tp->type, bufs = tp->bufs, flags = tp->flags;
    const uint32_t succ0 = tp->succ0, nsucc = tp->nsucc;
    if (L.trace && tid == 0) {
        L.trace[4 * (size_t)ticket] = t_start;
        L.trace[4 * (size_t)ticket + 1] = wall_clock64();
    }
    auto buf = [&](int b) -> double* { return b == 0 ? L.buf0 : (b == 1 ? L.buf1 : (b == 2 ? L.buf2 : L.buf3)); };
    auto ldof = [&](int b) -> int { return b == 0 ? L.ld0 : (b == 1 ? L.ld1 : (b == 2 ? L.ld2 : L.ld3)); };
    const int ab = bufs & 3, bb = (bufs >> 2) & 3, cb = (bufs >> 4) & 3;
    if (type == DAG_LEAF) {
        double* const b0 = lds0 + 256;
        double* const b1 = lds1;
        double* const b2 = lds2;
        leaf::potrf_trtri_tile<4>(buf(ab) + a_off, ldof(ab), buf(cb) + c_off, ldof(cb), kb * 128, L.info, lds0, [=](int bi, int bj) {
            const int idx = bi * (bi + 1) / 2 + bj;
            return idx < 16 ? b0 + idx * leaf::BS : (idx < 33 ? b1 + (idx - 16) * leaf::BS : b2 + (idx - 33) * leaf::BS);
        });
    } else {
        const double* A = buf(ab) + a_off;
        const double* B = buf(bb) + b_off;
        double* C = buf(cb) + c_off;
        const int lda = ldof(ab), ldb = ldof(bb), ldc = ldof(cb);
        const double alpha = (flags & DAG_ALPHA_NEG) ? -1.0 : 1.0, beta = (flags & DAG_BETA_ONE) ? 1.0 : 0.0;
        const bool mirror = (flags & DAG_MIRROR) != 0, down = (flags & DAG_DOWN) != 0;
        switch (type) {
            case DAG_GEMM_NT: dma_tile_product<false, false, 4>(A, lda, B, ldb, C, ldc, it * 128, jt * 128, kb * 128, ke * 128, down, alpha, beta, mirror, lds0, lds1); break;
            case DAG_GEMM_NN: dma_tile_product<false, true, 4>(A, lda, B, ldb, C, ldc, it * 128, jt * 128, kb * 128, ke * 128, down, alpha, beta, mirror, lds0, lds1); break;
            case DAG_GEMM_TN: dma_tile_product<true, true, 4>(A, lda, B, ldb, C, ldc, it * 128, jt * 128, kb * 128, ke * 128, down, alpha, beta, mirror, lds0, lds1); break;
            case DAG_GEMM_NT | DAG_TILE64: reg_tile_product<false, false, 64, 4>(A, lda, B, ldb, C, ldc, it * 64, jt * 64, kb * 128, ke * 128, alpha, beta, mirror, lds0, lds1); break;
            case DAG_GEMM_NN | DAG_TILE64: reg_tile_product<false, true, 64, 4>(A, lda, B, ldb, C, ldc, it * 64, jt * 64, kb * 128, ke * 128, alpha, beta, mirror, lds0, lds1); break;
            default: reg_tile_product<true, true, 64, 4>(A, lda, B, ldb, C, ldc, it * 64, jt * 64, kb * 128, ke * 128, alpha, beta, mirror, lds0, lds1); break;
        }
    }

    // ---- publish: every wave's stores have left, then ONE release at agent scope, then the successors hear of it ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 64) {
        if (L.trace && tid == 0) L.trace[4 * (size_t)ticket + 3] = wall_clock64();      // the task's own work ends here
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Sixteen runs at a time (a run has at most 64 tasks: one per lane): the decrements of all of them are on their way before
        // the first answer is looked at -- one round trip to the counters per sixteen runs, not per run
        uint32_t* const pending = L.state + L.state_pending;
        for (uint32_t r0 = 0; r0 < nsucc; r0 += 16) {
            uint32_t was[16], ids[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                was[j] = 0;
                ids[j] = 0;
                if (r0 + j < nsucc) {
                    const DagRun R = L.succ[succ0 + r0 + j];
                    if ((uint32_t)tid < R.count) {
                        ids[j] = R.first + (uint32_t)tid * (uint32_t)R.stride;
                        was[j] = __hip_atomic_fetch_sub(pending + ids[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (was[j] != 1u) continue;
                // the last predecessor: the successor is ready
                const uint32_t t = L.id2task[ids[j]];
                if (__hip_atomic_fetch_add((int*)(L.state + DAG_STATE_BALANCE), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) {
                    // a workgroup is waiting: the task goes straight to the next unserved one's mailbox
                    const uint32_t hnd = __hip_atomic_fetch_add(L.state + DAG_STATE_HANDOFFS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(L.state + L.state_mail + hnd, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const uint32_t q = L.tasks[t].queue;
                    const uint32_t pos = __hip_atomic_fetch_add(tails + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(L.state + L.state_slots + L.state[2 * DAG_QUEUES + q] + pos, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (L.trace && tid == 0) L.trace[4 * (size_t)ticket + 2] = wall_clock64();
    }
}

void launch_tile_dag(const DagLaunch& L, hipStream_t s) {
    if (!L.ntasks) return;
    hipLaunchKernelGGL(tile_dag_kernel, dim3(L.ntasks), dim3(256), 0, s, L);
}

// ------------------------------------------------------------------------------------------------------------------------------
// CPU execution (tests)
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

void host_task(const DagTask& t, double* const buf[DAG_MAX_BUFS], const int ld[DAG_MAX_BUFS]) {
    const int ab = t.bufs & 3, bb = (t.bufs >> 2) & 3, cb = (t.bufs >> 4) & 3;
    if (t.type == DAG_LEAF) {
        const double* A = buf[ab] + t.a_off;
        double* X = buf[cb] + t.c_off;
        const int lda = ld[ab], ldx = ld[cb];
        static thread_local std::vector<double> Ls(128 * 128);
        double* Lm = Ls.data();
        for (int j = 0; j < 128; ++j)
            for (int i = 0; i < 128; ++i) Lm[j * 128 + i] = i >= j ? A[(size_t)j * lda + i] : 0.0;
        for (int k = 0; k < 128; ++k) {
            double p = Lm[k * 128 + k];
            if (!(p > 0.0)) p = 1.0;
            const double d = std::sqrt(p);
            Lm[k * 128 + k] = d;
            for (int i = k + 1; i < 128; ++i) Lm[k * 128 + i] /= d;
            for (int j = k + 1; j < 128; ++j)
                for (int i = j; i < 128; ++i) Lm[j * 128 + i] -= Lm[k * 128 + i] * Lm[k * 128 + j];
        }
        for (int j = 0; j < 128; ++j) {          // column j of L^-1 by forward substitution
            for (int i = 0; i < 128; ++i) {
                double s = i == j ? 1.0 : 0.0;
                if (i < j) {
                    X[(size_t)j * ldx + i] = 0.0;
                    continue;
                }
                for (int k = j; k < i; ++k) s -= Lm[k * 128 + i] * X[(size_t)j * ldx + k];
                X[(size_t)j * ldx + i] = s / Lm[i * 128 + i];
            }
        }
        return;
    }
    const double* A = buf[ab] + t.a_off;
    const double* B = buf[bb] + t.b_off;
    double* C = buf[cb] + t.c_off;
    const int lda = ld[ab], ldb = ld[bb], ldc = ld[cb];
    const bool akc = (t.type & 3) == DAG_GEMM_TN, bkc = (t.type & 3) != DAG_GEMM_NT;
    const double alpha = (t.flags & DAG_ALPHA_NEG) ? -1.0 : 1.0;
    const bool beta = (t.flags & DAG_BETA_ONE) != 0;
    const int TS = (t.type & DAG_TILE64) ? 64 : 128;
    const int i0 = t.it * TS, j0 = t.jt * TS, k0 = t.kb * 128, k1 = t.ke * 128;
    static thread_local std::vector<double> acc(128 * 128);
    std::fill(acc.begin(), acc.end(), 0.0);
    for (int k = k0; k < k1; ++k)
        for (int j = 0; j < TS; ++j) {
            const double b = bkc ? B[(size_t)(j0 + j) * ldb + k] : B[(size_t)k * ldb + j0 + j];
            for (int i = 0; i < TS; ++i) {
                const double a = akc ? A[(size_t)(i0 + i) * lda + k] : A[(size_t)k * lda + i0 + i];
                acc[j * 128 + i] += a * b;
            }
        }
    for (int j = 0; j < TS; ++j)
        for (int i = 0; i < TS; ++i) {
            double v = alpha * acc[j * 128 + i];
            if (beta) v += C[(size_t)(j0 + j) * ldc + i0 + i];
            acc[j * 128 + i] = v;
        }
    for (int j = 0; j < TS; ++j)
        for (int i = 0; i < TS; ++i) {
            C[(size_t)(j0 + j) * ldc + i0 + i] = acc[j * 128 + i];
            if (t.flags & DAG_MIRROR) C[(size_t)(i0 + i) * ldc + j0 + j] = acc[j * 128 + i];
        }
}

}  // namespace

bool dag_execute_host(const DagGraph& g, double* const buf[DAG_MAX_BUFS], const int ld[DAG_MAX_BUFS], int order, uint64_t seed) {
    const size_t N = g.tasks.size();
    std::vector<uint32_t> st(g.state_init);                  // the device's state words, used the way the kernel uses them
    uint32_t* heads = st.data();
    uint32_t* tails = st.data() + DAG_QUEUES;
    const uint32_t* slot_base = st.data() + 2 * DAG_QUEUES;
    uint32_t* pending = st.data() + g.state_pending;
    uint32_t* slots = st.data() + g.state_slots;
    std::vector<uint8_t> ran(N, 0);
    uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    auto next = [&] {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        return rng;
    };
    std::vector<uint32_t> by_id(N);                          // tasks in recorded order
    for (size_t k = 0; k < N; ++k) by_id[k] = (uint32_t)k;
    std::sort(by_id.begin(), by_id.end(), [&](uint32_t x, uint32_t y) { return g.tasks[x].id < g.tasks[y].id; });
    // ready tasks not yet run, as the queues know them (entries put and not taken)
    auto finish = [&](uint32_t k) -> bool {
        const DagTask& t = g.tasks[k];
        for (uint32_t r = 0; r < t.nsucc; ++r) {
            const DagRun& R = g.succ[t.succ0 + r];
            for (uint32_t c = 0; c < R.count; ++c) {
                const uint32_t id = R.first + c * R.stride;
                if (id >= g.nids || pending[id] == 0) return false;
                if (--pending[id] == 0) {
                    const uint32_t s = g.id2task[id];
                    if (s == 0xffffffffu) return false;
                    const uint32_t q = g.tasks[s].queue;
                    if (slot_base[q] + tails[q] >= slot_base[q + 1]) return false;
                    slots[slot_base[q] + tails[q]++] = s + 1;
                }
            }
        }
        return true;
    };
    for (size_t n_run = 0; n_run < N; ++n_run) {
        uint32_t pick = 0xffffffffu;
        if (order == 3) {
            for (int q = 0; q < DAG_QUEUES && pick == 0xffffffffu; ++q)
                if (heads[q] < tails[q]) pick = slots[slot_base[q] + heads[q]++] - 1;
        } else {
            // every entry between a queue's head and tail that has not run yet is ready
            std::vector<uint32_t> rd;
            for (int q = 0; q < DAG_QUEUES; ++q)
                for (uint32_t p = 0; p < tails[q]; ++p) {
                    const uint32_t k = slots[slot_base[q] + p] - 1;
                    if (!ran[k]) rd.push_back(k);
                }
            if (order == 0) {
                pick = by_id[n_run];
                if (std::find(rd.begin(), rd.end(), pick) == rd.end()) return false;      // the recorded order itself violates a counter
            } else if (!rd.empty()) {
                if (order == 2) {
                    pick = rd[0];
                    for (uint32_t k : rd)
                        if (g.tasks[k].id > g.tasks[pick].id) pick = k;
                } else {
                    pick = rd[next() % rd.size()];
                }
            }
        }
        if (pick == 0xffffffffu || pick >= N || ran[pick]) return false;       // a stall: tasks left, none ready
        host_task(g.tasks[pick], buf, ld);
        ran[pick] = 1;
        if (!finish(pick)) return false;
    }
    for (size_t i = 0; i < g.nids; ++i)
        if (pending[i]) return false;
    return true;
}

}  // namespace dnagpu
