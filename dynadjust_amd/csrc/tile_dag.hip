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
constexpr float DAG_HOP_US = 6.0f;

// recorded numbers (sorted, unique) -> arithmetic runs
void compress(const std::vector<uint32_t>& p, std::vector<DagRun>& out) {
    size_t i = 0;
    while (i < p.size()) {
        DagRun d{p[i], 1, 1};
        if (i + 1 < p.size()) {
            const uint32_t stride = p[i + 1] - p[i];
            if (stride <= 0xffffu) {
                size_t j = i + 1;
                while (j < p.size() && p[j] - p[j - 1] == stride && d.count < 64) {       // (one flag per lane of the wave that checks the run)
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

std::shared_ptr<DagGraph> DagBuilder::finish(int reorder, int workers) {
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
    // predecessors as runs of recorded numbers (what the device waits for)
    for (size_t i = 0; i < N; ++i) {
        preds.clear();
        for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) preds.push_back(T[pred_idx[q]].id);
        std::sort(preds.begin(), preds.end());
        T[i].dep0 = (uint32_t)g->deps.size();
        compress(preds, g->deps);
        T[i].ndep = (uint32_t)g->deps.size() - T[i].dep0;
    }

    // ---- launch order: list scheduling with critical-path priorities ----
    // a successor starts DAG_HOP_US after its last predecessor ended: release, flag, poll, acquire
    std::vector<float> dur(N), bl(N);
    for (size_t i = 0; i < N; ++i) {
        dur[i] = task_us(T[i]) + DAG_HOP_US;
        bl[i] = dur[i];
        g->sim_work_us += task_us(T[i]);
    }
    for (size_t i = N; i-- > 0;)
        for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) {
            const uint32_t p = pred_idx[q];
            bl[p] = std::max(bl[p], dur[p] + bl[i]);
        }
    for (size_t i = 0; i < N; ++i) g->critical_path_us = std::max(g->critical_path_us, (double)bl[i]);
    std::vector<uint32_t> order(N);
    for (size_t i = 0; i < N; ++i) order[i] = (uint32_t)i;
    if (reorder && N > 1) {
        std::vector<uint32_t> succ_off(N + 1, 0), succ_idx(pred_idx.size()), indeg(N);
        for (size_t i = 0; i < N; ++i) {
            indeg[i] = pred_off[i + 1] - pred_off[i];
            for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) succ_off[pred_idx[q] + 1]++;
        }
        for (size_t i = 0; i < N; ++i) succ_off[i + 1] += succ_off[i];
        {
            std::vector<uint32_t> fill(succ_off.begin(), succ_off.end() - 1);
            for (size_t i = 0; i < N; ++i)
                for (uint32_t q = pred_off[i]; q < pred_off[i + 1]; ++q) succ_idx[fill[pred_idx[q]]++] = (uint32_t)i;
        }
        struct Ready {
            float bl;
            uint32_t i;
            bool operator<(const Ready& o) const { return bl != o.bl ? bl < o.bl : i > o.i; }     // max-heap: longest path first, then recorded order
        };
        std::priority_queue<Ready> ready;
        typedef std::pair<double, uint32_t> Fin;
        std::priority_queue<Fin, std::vector<Fin>, std::greater<Fin>> running;
        for (size_t i = 0; i < N; ++i)
            if (!indeg[i]) ready.push({bl[i], (uint32_t)i});
        order.clear();
        order.reserve(N);
        double now = 0.0;
        int free_w = std::max(1, workers);
        while (order.size() < N) {
            while (free_w > 0 && !ready.empty()) {
                const uint32_t i = ready.top().i;
                ready.pop();
                order.push_back(i);
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
                    if (--indeg[succ_idx[q]] == 0) ready.push({bl[succ_idx[q]], succ_idx[q]});
            }
        }
        while (!running.empty()) {
            now = running.top().first;
            running.pop();
        }
        g->sim_makespan_us = now;
        if (order.size() != N) {
            order.resize(N);
            for (size_t i = 0; i < N; ++i) order[i] = (uint32_t)i;
        }
    }
    g->tasks.resize(N);
    for (size_t k = 0; k < N; ++k) g->tasks[k] = T[order[k]];
    return g;
}

DagGraph::~DagGraph() {
    if (d_tasks || d_deps) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (device >= 0 && device != cur) (void)hipSetDevice(device);
        if (d_tasks) (void)hipFree(d_tasks);
        if (d_deps) (void)hipFree(d_deps);
        if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    }
}

hipError_t dag_upload(DagGraph& g) {
    if (g.d_tasks) return hipSuccess;
    hipError_t e = hipGetDevice(&g.device);
    if (e != hipSuccess) return e;
    if ((e = hipMalloc(&g.d_tasks, std::max<size_t>(1, g.tasks.size()) * sizeof(DagTask))) != hipSuccess) return e;
    if ((e = hipMalloc(&g.d_deps, std::max<size_t>(1, g.deps.size()) * sizeof(DagRun))) != hipSuccess) return e;
    if (!g.tasks.empty() && (e = hipMemcpy(g.d_tasks, g.tasks.data(), g.tasks.size() * sizeof(DagTask), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if (!g.deps.empty() && (e = hipMemcpy(g.d_deps, g.deps.data(), g.deps.size() * sizeof(DagRun), hipMemcpyHostToDevice)) != hipSuccess) return e;
    return hipSuccess;
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
    __shared__ uint32_t s_ticket;
    static_assert(2 * G::OPBUF == 256 + 16 * leaf::BS && 2 * G::OPBUF + 16 == 17 * leaf::BS, "leaf blocks in the operand buffers");
    const int tid = threadIdx.x;
    // A persistent worker: one task per round, until the tickets run out.  Thread 0 takes the next ticket in the SAME block in which it
    // publishes the finished task, at the end of the round, and a round begins with the barrier: with the two thread-0 blocks on either
    // side of the loop's back edge the compiler fused them and sent the other lanes round an inner loop of their own -- the waves then
    // met the barriers a different number of times (a hang on the device; nothing a one-task-per-workgroup kernel would ever show).
    if (tid == 0) s_ticket = (uint32_t)(atomicAdd(L.ticket, 1ULL) - L.ticket_base);
    for (;;) {
    __syncthreads();
    const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);
    if (ticket >= L.ntasks) return;
    // the task's nine words in ONE round trip (lane l fetches word l), broadcast by readlane
    static_assert(sizeof(DagTask) == 36, "nine words");
    uint32_t word = 0;
    if ((tid & 63) < 9) word = reinterpret_cast<const uint32_t*>(L.tasks + ticket)[tid & 63];
    auto field = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)word, i); };
    const uint32_t a_off = field(0), b_off = field(1), c_off = field(2);
    const int it = (int)(field(3) & 0xffffu), jt = (int)(field(3) >> 16), kb = (int)(field(4) & 0xffffu), ke = (int)(field(4) >> 16);
    const int type = (int)(field(5) & 0xffu), bufs = (int)((field(5) >> 8) & 0xffu), flags = (int)((field(5) >> 16) & 0xffu);
    const uint32_t my_flag = field(6), dep0 = field(7), ndep = field(8);
    if (L.trace && tid == 0) L.trace[4 * (size_t)ticket] = wall_clock64();

    // ---- wait for the predecessors (wave 0).  A run has at most 64 flags: one per lane.  The runs' descriptors come 64 at a time (lane l
    // fetches run l), the flags of eight runs are on their way before the first is looked at: a check of everything costs two or
    // three round trips, not two per run -- on the critical path that is the larger part of a hop. ----
    if (ndep) {
        if (tid < 64) {
            long spins = 0;
            for (;;) {
                bool ok = true;
                for (uint32_t r0 = 0; r0 < ndep; r0 += 64) {
                    uint32_t first = 0, cs = 0;                              // count | stride << 16
                    if (r0 + (uint32_t)tid < ndep) {
                        const uint32_t* d = reinterpret_cast<const uint32_t*>(L.deps + dep0 + r0 + tid);
                        first = d[0];
                        cs = d[1];
                    }
                    const uint32_t nr = ndep - r0 < 64u ? ndep - r0 : 64u;
                    for (uint32_t j0 = 0; j0 < nr; j0 += 8) {
                        uint32_t v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            v[j] = L.epoch;
                            if (j0 + j < nr) {
                                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)first, j0 + j);
                                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cs, j0 + j);
                                if ((uint32_t)tid < (c & 0xffffu))
                                    v[j] = __hip_atomic_load(L.flags + f + (uint32_t)tid * (c >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) ok = ok && v[j] == L.epoch;
                    }
                }
                if (!__ballot(!ok)) break;
                __builtin_amdgcn_s_sleep(8);
                ++spins;
                // seconds without a flag: a predecessor never finished.  The first worker to give up says so in `info`, the others see
                // that and stop waiting too: the launch ends (with a void result) instead of hanging
                if (spins > (1L << 21) || ((spins & 255) == 0 && __hip_atomic_load(L.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == INFO_BARRIER_TIMEOUT)) {
                    if (tid == 0) atomicMin(L.info, INFO_BARRIER_TIMEOUT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // this CU's L1 forgets what it held of the predecessors' tiles
        }
        __syncthreads();
        if (L.paranoid) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (diagnostic: every wave, after the barrier)
    }
    if (L.trace && tid == 0) L.trace[4 * (size_t)ticket + 1] = wall_clock64();
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

    // ---- publish: every wave's stores have left, then ONE release at agent scope, then the flag ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                             // (also: every wave is done with the LDS before the next task's operands land in it)
    if (tid == 0) {
        if (L.trace) L.trace[4 * (size_t)ticket + 3] = wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(L.flags + my_flag, L.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (L.trace) L.trace[4 * (size_t)ticket + 2] = wall_clock64();
        s_ticket = (uint32_t)(atomicAdd(L.ticket, 1ULL) - L.ticket_base);      // (the others read the old one before the barrier above)
    }
    }   // worker loop
}

void launch_tile_dag(const DagLaunch& L, hipStream_t s) {
    if (!L.ntasks || L.workers <= 0) return;
    hipLaunchKernelGGL(tile_dag_kernel, dim3(L.workers), dim3(256), 0, s, L);
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
    std::vector<uint8_t> done(g.nids, 0), ran(N, 0);
    auto is_ready = [&](const DagTask& t) {
        for (uint32_t d = 0; d < t.ndep; ++d) {
            const DagRun& D = g.deps[t.dep0 + d];
            for (uint32_t c = 0; c < D.count; ++c)
                if (!done[D.first + c * D.stride]) return false;
        }
        return true;
    };
    uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    auto next = [&] {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        return rng;
    };
    size_t n_run = 0;
    while (n_run < N) {
        size_t pick = N;
        if (order == 0) {
            pick = n_run;
            if (!is_ready(g.tasks[pick])) return false;       // the list is not a topological order of its own flags
        } else if (order == 2) {
            for (size_t i = N; i-- > 0;)
                if (!ran[i] && is_ready(g.tasks[i])) {
                    pick = i;
                    break;
                }
        } else {
            std::vector<size_t> rd;
            if (next() & 1) {
                for (size_t i = 0; i < N && rd.size() < 64; ++i)
                    if (!ran[i] && is_ready(g.tasks[i])) rd.push_back(i);
            } else {
                for (size_t i = N; i-- > 0 && rd.size() < 64;)
                    if (!ran[i] && is_ready(g.tasks[i])) rd.push_back(i);
            }
            if (!rd.empty()) pick = rd[next() % rd.size()];
        }
        if (pick == N) return false;                            // nothing ready: a cycle
        host_task(g.tasks[pick], buf, ld);
        ran[pick] = 1;
        done[g.tasks[pick].id] = 1;
        ++n_run;
    }
    return true;
}

}  // namespace dnagpu
