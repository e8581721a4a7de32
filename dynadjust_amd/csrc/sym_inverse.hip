#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <cstring>
#include <cstdio>
#include "sym_inverse.h"
#include "la_kernels.h"
#include "tile_dag.h"
#include <algorithm>
#include <cstdlib>
#include <atomic>

namespace dnagpu {

hipError_t inv_workspace_alloc(InvWorkspace& ws, uint32_t np_cap, hipStream_t stream) {
    ws.np_cap = np_cap;
    ws.stream = stream;
    size_t bytes = (size_t)np_cap * np_cap * sizeof(double);
    hipError_t e;
    if ((e = dnagpu::poison_malloc(&ws.X, bytes)) != hipSuccess) return e;
    if ((e = dnagpu::poison_malloc(&ws.W, bytes)) != hipSuccess) return e;
    if ((e = dnagpu::poison_malloc(&ws.svec, (size_t)np_cap * sizeof(double))) != hipSuccess) return e;
    if ((e = dnagpu::poison_malloc(&ws.info, BATCH_MAX * sizeof(int))) != hipSuccess) return e;
    if ((e = dnagpu::poison_malloc(&ws.sync_ctr, sizeof(unsigned long long))) != hipSuccess) return e;
    if ((e = hipMemset(ws.sync_ctr, 0, sizeof(unsigned long long))) != hipSuccess) return e;
    ws.sync_base = 0;
    if ((e = hipHostMalloc(&ws.info_host, BATCH_MAX * sizeof(int))) != hipSuccess) return e;
    for (int b = 0; b < BATCH_MAX; ++b) ws.info_host[b] = 0;
    return hipSuccess;
}

void inv_workspace_free(InvWorkspace& ws) {
    if (ws.dag_flags) hipFree(ws.dag_flags);
    if (ws.dag_ticket) hipFree(ws.dag_ticket);
    if (ws.X) hipFree(ws.X);
    if (ws.W) hipFree(ws.W);
    for (int b = 0; b < BATCH_MAX; ++b) {
        if (ws.bX[b]) hipFree(ws.bX[b]);
        if (ws.bW[b]) hipFree(ws.bW[b]);
    }
    if (ws.svec) hipFree(ws.svec);
    if (ws.info) hipFree(ws.info);
    if (ws.sync_ctr) hipFree(ws.sync_ctr);
    if (ws.dist_stage) hipFree(ws.dist_stage);
    if (ws.info_host) hipHostFree(ws.info_host);
    for (hipEvent_t ev : ws.prof.pool) hipEventDestroy(ev);
    for (hipEvent_t ev : ws.prof.side_pool) hipEventDestroy(ev);
    for (hipEvent_t ev : ws.la_events) hipEventDestroy(ev);
    for (hipStream_t st : ws.side)
        if (st) hipStreamDestroy(st);
    for (auto& kv : ws.order_cache)
        if (kv.second.first) hipFree(kv.second.first);
    ws = InvWorkspace();
}

void inv_note_error(InvWorkspace& ws, hipError_t e, const char* where) {
    if (e != hipSuccess && ws.err == hipSuccess) {
        ws.err = e;
        ws.err_where = where;
    }
}

hipError_t inv_take_error(InvWorkspace& ws, const char** where) {
    hipError_t e = ws.err;
    if (where) *where = ws.err_where;
    ws.err = hipSuccess;
    ws.err_where = nullptr;
    return e;
}

// fault injection for the error-path tests: countdown to a failing table allocation
static std::atomic<long> g_fault_countdown{[] {
    const char* e = getenv("DNAGPU_FAULT_INJECT");
    return e ? atol(e) : 0L;
}()};
void fault_inject_reset(long nth) { g_fault_countdown.store(nth); }
static hipError_t table_malloc(uint32_t** dev, size_t bytes) {
    if (g_fault_countdown.load() > 0 && g_fault_countdown.fetch_sub(1) == 1) return hipErrorOutOfMemory;
    return dnagpu::poison_malloc(dev, bytes);
}

static double gemm_flops(const GemmArgs& a) {
    // sum over launched tiles of 2*128*128*klen(tile)
    double f = 0.0;
    for (int it = 0; it < a.mt; ++it) {
        int jmax = a.lower ? it : a.nt - 1;
        for (int jt = 0; jt <= jmax; ++jt) {
            int kb = 0, ke = a.K;
            switch (a.kmode) {
                case KM_LE_J: ke = (jt + 1) * 128; break;
                case KM_GE_J: kb = jt * 128; break;
                case KM_LE_I: ke = (it + 1) * 128; break;
                case KM_GE_I: kb = it * 128; break;
                default: break;
            }
            if (ke > a.K) ke = a.K;
            if (ke > kb) f += 2.0 * 128.0 * 128.0 * (double)(ke - kb);
        }
    }
    return f;
}

// launches with fewer 128-tiles than this run on 64 x 64 block tiles (DNAGPU_SMALL_TILES, dnagpu_debug_set_small_tiles: 0 sends
// every launch through the 128-tile throughput kernel -- how the tests compare THAT kernel with the oracle at small orders)
static std::atomic<long> g_small_tiles{[] {
    const char* e = getenv("DNAGPU_SMALL_TILES");
    return e ? atol(e) : (long)SMALL_LAUNCH_TILES;
}()};
static std::atomic<long> g_tiny_tiles{[] {
    const char* e = getenv("DNAGPU_TINY_TILES");
    return e ? atol(e) : (long)TINY_LAUNCH_TILES;
}()};
long tiny_tiles_set(long v) {
    long old = g_tiny_tiles.load();
    g_tiny_tiles.store(v < 0 ? (long)TINY_LAUNCH_TILES : v);
    return old;
}
long small_tiles_set(long v) {
    long old = g_small_tiles.load();
    g_small_tiles.store(v < 0 ? (long)SMALL_LAUNCH_TILES : v);
    return old;
}

hipError_t gemm_attach_order(InvWorkspace& ws, GemmArgs& a, int jt_lo, int jt_hi) {
    long total = a.lower ? (long)a.mt * (a.mt + 1) / 2 : (long)a.mt * a.nt;
    // (a batched launch decides by one member's tiles, like the unbatched launch whose bits it must reproduce; counting all
    //  members' tiles -- the 128-tile shape from fewer tiles per member on -- measured no different: 2 479 against 2 483 ms per cfg3 step)
    // (the opt-in fused launches walk 64-tiles: with them on, the 32-tile shape stays out)
    a.tile = total < g_small_tiles.load() ? ((total < g_tiny_tiles.load() && !ws.fuse) ? 32 : 64) : 128;
    // (the pair threshold is part of the key: a table built before dnagpu_debug_set_pair_tiles changed it is not the one wanted after)
    const long pair_from = pair_tiles_get();
    const uint64_t shape = (uint64_t)a.mt | ((uint64_t)a.nt << 16) | ((uint64_t)a.kmode << 32) | ((uint64_t)(a.lower ? 1 : 0) << 36) |
                           ((uint64_t)(a.tile == 64 ? 1 : 0) << 37) | ((uint64_t)(pair_from > 0 && total >= pair_from ? 1 : 0) << 38) | ((uint64_t)(a.tile == 32 ? 1 : 0) << 39) |
                           ((uint64_t)(a.K / 16) << 40);
    const auto key = std::make_pair(shape, jt_lo < 0 ? 0xffffffffu : ((uint32_t)jt_lo << 16) | (uint32_t)jt_hi);
    auto it = ws.order_cache.find(key);
    if (it == ws.order_cache.end()) {
        int pairs = 0;
        std::vector<uint32_t> tab = build_tile_order(a.mt, a.nt, a.K, a.kmode, a.lower, a.tile, jt_lo, jt_hi, &pairs);
        uint32_t* dev = nullptr;
        if (!tab.empty()) {
            hipError_t e = table_malloc(&dev, tab.size() * sizeof(uint32_t));
            if (e != hipSuccess) {
                (void)hipGetLastError();
                inv_note_error(ws, e, "tile-order table allocation");
                return e;
            }
            // synchronous copy: the table is immutable afterwards
            e = hipMemcpy(dev, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                hipFree(dev);
                inv_note_error(ws, e, "tile-order table upload");
                return e;
            }
        }
        // (a table with two entries per workgroup carries its grid negated)
        it = ws.order_cache.emplace(key, std::make_pair(dev, pairs ? -(int)(tab.size() / 2) : (int)tab.size())).first;
    }
    a.order = it->second.first;
    a.grid = it->second.second < 0 ? -it->second.second : it->second.second;
    a.pairs = it->second.second < 0 ? 1 : 0;
    return hipSuccess;
}

// HIP-event timing of the GEMM launches (bench.py's roofline): one event pair brackets every maximal RUN of
// consecutive gemm launches on the stream (a run ends when any other kernel is enqueued: gemm_profile_close).
// Bracketing every single launch cost 3 % of a cfg3 step (1 600 extra barrier packets per inverse); runs cost < 1 %.
static void profile_event(GemmProfile& p, hipStream_t s) {
    if (p.used + 1 > p.pool.size()) {
        hipEvent_t ev;
        hipEventCreate(&ev);
        p.pool.push_back(ev);
    }
    hipEventRecord(p.pool[p.used++], s);
}

void gemm_flush(InvWorkspace& ws) {
    if (ws.pending.empty()) return;
    if (ws.err == hipSuccess) {
        FusedArgs f;
        f.nops = (int)ws.pending.size();
        f.counter = ws.sync_ctr;
        f.base = ws.sync_base;
        f.info = ws.info;
        int grid = 1;
        for (int i = 0; i < f.nops; ++i) {
            f.op[i] = ws.pending[i];
            const long mt = 2L * f.op[i].mt, nt = 2L * f.op[i].nt;
            const long tiles = f.op[i].lower ? mt * (mt + 1) / 2 : mt * nt;
            grid = (int)std::max<long>(grid, tiles);
        }
        launch_gemm_fused(f, grid, ws.stream);
        inv_note_error(ws, hipGetLastError(), "fused GEMM launch");
        ws.sync_base += (unsigned long long)grid * (unsigned long long)(f.nops - 1);
        ws.fused_launches++;
        ws.fused_ops += (uint64_t)f.nops;
    }
    ws.pending.clear();
}

void gemm_fused_reset(InvWorkspace& ws) {
    ws.pending.clear();
    if (ws.sync_ctr) hipMemset(ws.sync_ctr, 0, sizeof(unsigned long long));
    ws.sync_base = 0;
}

void gemm_profile_close(InvWorkspace& ws) {
    gemm_flush(ws);
    GemmProfile& p = ws.prof;
    if (!p.open) return;
    profile_event(p, ws.stream);
    p.open = false;
}

// One large launch across the GPUs of an intra-block distributed inverse: every rank computes the tile columns of its range, packs
// them (the rows that the launch writes, column by column) into its part of the staging buffer, all parts travel to all ranks
// (ws.exchange: one broadcast per rank, on this stream), the others' parts are unpacked into the matrix.  A mirrored launch (LAUUM)
// writes its lower tiles only and the upper triangle is filled by every rank afterwards.
static bool gemm_split(InvWorkspace& ws, GemmArgs a, int akc, int bkc) {
    const int W = ws.dist_world, me = ws.dist_rank;
    const long total = a.lower ? (long)a.mt * (a.mt + 1) / 2 : (long)a.mt * a.nt;
    if (W < 2 || !ws.exchange || total < g_small_tiles.load()) return false;
    gemm_flush(ws);
    const std::vector<int> lo = split_tile_columns(a.mt, a.nt, a.K, a.kmode, a.lower, W);
    const bool mirror = a.mirror != 0;
    a.mirror = 0;
    // the parts: rows [row0, mt * 128) of columns [lo[q], lo[q + 1]) * 128
    std::vector<size_t> off(W + 1, 0), counts(W), rows(W), row0(W), cols(W);
    for (int q = 0; q < W; ++q) {
        row0[q] = a.lower ? (size_t)lo[q] * 128 : 0;
        rows[q] = (size_t)a.mt * 128 - row0[q];
        cols[q] = (size_t)(lo[q + 1] - lo[q]) * 128;
        counts[q] = rows[q] * cols[q];
        off[q + 1] = off[q] + counts[q];
    }
    if (ws.dist_stage_cap < off[W]) {
        hipStreamSynchronize(ws.stream);
        if (ws.dist_stage) hipFree(ws.dist_stage);
        ws.dist_stage = nullptr;
        ws.dist_stage_cap = 0;
        const size_t want = std::max(off[W], (size_t)ws.np_cap * ws.np_cap);
        hipError_t e = dnagpu::poison_malloc(&ws.dist_stage, want * sizeof(double));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            inv_note_error(ws, e, "staging buffer of the distributed inverse");
            return true;
        }
        ws.dist_stage_cap = want;
    }
    if (ws.err != hipSuccess) return true;
    if (lo[me + 1] > lo[me]) {
        GemmArgs mine = a;
        if (gemm_attach_order(ws, mine, lo[me], lo[me + 1]) != hipSuccess) return true;
        GemmProfile& p = ws.prof;
        if (p.enabled) {
            if (!p.open) {
                profile_event(p, ws.stream);
                p.open = true;
            }
            p.flops += gemm_flops(a) / W;
            p.launches++;
        }
        launch_gemm(mine, akc, bkc, ws.stream);
        inv_note_error(ws, hipGetLastError(), "tile GEMM launch");
    }
    gemm_profile_close(ws);
    auto region = [&](int q) { return a.C + (size_t)lo[q] * 128 * a.ldc + row0[q]; };
    if (counts[me])
        inv_note_error(ws, hipMemcpy2DAsync(ws.dist_stage + off[me], rows[me] * sizeof(double), region(me), (size_t)a.ldc * sizeof(double),
                                            rows[me] * sizeof(double), cols[me], hipMemcpyDeviceToDevice, ws.stream), "pack");
    std::vector<double*> bufs(W);
    for (int q = 0; q < W; ++q) bufs[q] = ws.dist_stage + off[q];
    if (ws.err == hipSuccess && ws.exchange(ws.exchange_user, (void*)ws.stream, W, bufs.data(), counts.data()) != 0)
        inv_note_error(ws, hipErrorUnknown, "exchange of the distributed inverse");
    for (int q = 0; q < W && ws.err == hipSuccess; ++q)
        if (q != me && counts[q])
            inv_note_error(ws, hipMemcpy2DAsync(region(q), (size_t)a.ldc * sizeof(double), ws.dist_stage + off[q], rows[q] * sizeof(double),
                                                rows[q] * sizeof(double), cols[q], hipMemcpyDeviceToDevice, ws.stream), "unpack");
    if (mirror && ws.err == hipSuccess) launch_symmetrize(a.C, (uint32_t)a.mt * 128, (uint32_t)a.ldc, ws.stream);
    ws.split_launches++;
    ws.exchanged_bytes += (double)(off[W] - counts[me]) * sizeof(double);
    return true;
}

// the offsets of the batch's members for an operand of member 0 (by the registered buffer its address lies in)
static bool batch_offsets(const InvBatch& bt, const double* p, long long* d) {
    for (int q = 0; q < bt.nbuf; ++q)
        if (p >= bt.base[q] && p < bt.base[q] + bt.span[q]) {
            for (int b = 0; b < bt.nb; ++b) d[b] = bt.delta[q][b];
            return true;
        }
    return false;
}

// One column range [jt_lo, jt_hi) of a lower-triangular launch, on `on` (the chain's stream, or one of its look-ahead streams):
// the same tiles, computed the same way, as in the whole launch.  count: this part carries the launch's flops in the profile.
void gemm_part(InvWorkspace& ws, GemmArgs a, int akc, int bkc, hipStream_t on, int jt_lo, int jt_hi, bool count) {
    if (ws.err != hipSuccess || gemm_attach_order(ws, a, jt_lo, jt_hi) != hipSuccess) return;
    if (ws.batch.nb > 1) {
        a.nb = ws.batch.nb;
        if (!batch_offsets(ws.batch, a.A, a.dA) || !batch_offsets(ws.batch, a.B, a.dB) || !batch_offsets(ws.batch, a.C, a.dC)) {
            inv_note_error(ws, hipErrorInvalidValue, "batched product: an operand outside the registered buffers");
            return;
        }
        ws.batched_launches++;
    }
    GemmProfile& p = ws.prof;
    const bool side = on != ws.stream;
    auto side_event = [&] {
        if (p.side_used + 1 > p.side_pool.size()) {
            hipEvent_t ev;
            hipEventCreate(&ev);
            p.side_pool.push_back(ev);
        }
        hipEventRecord(p.side_pool[p.side_used++], on);
    };
    if (p.enabled) {
        if (side) {
            side_event();
        } else if (!p.open) {
            profile_event(p, ws.stream);
            p.open = true;
        }
        if (count) {
            p.flops += gemm_flops(a) * a.nb;
            p.launches++;
        }
    }
    launch_gemm(a, akc, bkc, on);
    inv_note_error(ws, hipGetLastError(), "tile GEMM launch");
    if (p.enabled && side) side_event();
    if (side) ws.la_launches++;
}

void gemm(InvWorkspace& ws, GemmArgs a, int akc, int bkc) {
    if (ws.batch.nb > 1) {
        if (ws.err != hipSuccess || gemm_attach_order(ws, a) != hipSuccess) return;
        a.nb = ws.batch.nb;
        if (!batch_offsets(ws.batch, a.A, a.dA) || !batch_offsets(ws.batch, a.B, a.dB) || !batch_offsets(ws.batch, a.C, a.dC)) {
            inv_note_error(ws, hipErrorInvalidValue, "batched product: an operand outside the registered buffers");
            return;
        }
        GemmProfile& p = ws.prof;
        if (p.enabled) {
            if (!p.open) {
                profile_event(p, ws.stream);
                p.open = true;
            }
            p.flops += gemm_flops(a) * a.nb;
            p.launches++;
        }
        launch_gemm(a, akc, bkc, ws.stream);
        inv_note_error(ws, hipGetLastError(), "tile GEMM launch");
        ws.batched_launches++;
        return;
    }
    if (ws.dist_world > 1 && ws.err == hipSuccess && gemm_split(ws, a, akc, bkc)) return;
    // (after a latched error nothing further is enqueued: the result is void anyway and the caller reports the error)
    if (ws.err != hipSuccess || gemm_attach_order(ws, a) != hipSuccess) return;
    GemmProfile& p = ws.prof;
    if (p.enabled) {
        if (!p.open) {
            profile_event(p, ws.stream);
            p.open = true;
        }
        p.flops += gemm_flops(a);
        p.launches++;
    }
    // DNAGPU_GEMM_HISTOGRAM=1: launch shapes (tile, workgroups, K) counted and printed when the process ends (diagnostic)
    static const bool hist = getenv("DNAGPU_GEMM_HISTOGRAM") != nullptr;
    if (hist) {
        struct Hist {
            std::mutex m;
            std::map<std::tuple<int, int, int>, long> n;
            ~Hist() {
                for (auto& kv : n)
                    fprintf(stderr, "gemm tile=%d workgroups=%d K=%d launches=%ld\n", std::get<0>(kv.first), std::get<1>(kv.first),
                            std::get<2>(kv.first), kv.second);
            }
        };
        static Hist h;
        std::lock_guard<std::mutex> g(h.m);
        h.n[std::make_tuple(a.tile, a.grid, a.K)]++;
    }
    const long tiles64 = a.lower ? (2L * a.mt) * (2L * a.mt + 1) / 2 : 4L * a.mt * a.nt;
    if (a.tile == 64 && tiles64 <= FUSED_MAX_GRID && ws.fuse && ws.sync_ctr) {
        // a small product: waits for its neighbours (gemm_flush sends the run out as one launch)
        FusedOp op;
        op.A = a.A; op.B = a.B; op.C = a.C;
        op.lda = a.lda; op.ldb = a.ldb; op.ldc = a.ldc;
        op.mt = a.mt; op.nt = a.nt; op.K = a.K;
        op.alpha = a.alpha; op.beta = a.beta;
        op.kmode = a.kmode; op.lower = a.lower; op.mirror = a.mirror;
        op.akc = akc; op.bkc = bkc;
        ws.pending.push_back(op);
        if (ws.pending.size() >= (size_t)FUSED_MAX_OPS) gemm_flush(ws);
        return;
    }
    gemm_flush(ws);
    launch_gemm(a, akc, bkc, ws.stream);
    inv_note_error(ws, hipGetLastError(), "tile GEMM launch");
}

void gemm_profile_collect(InvWorkspace& ws) {
    GemmProfile& p = ws.prof;
    gemm_profile_close(ws);
    hipStreamSynchronize(ws.stream);
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, p.pool[i], p.pool[i + 1]);
        p.gemm_ms += ms;
    }
    p.used = 0;
    p.side_used = 0;        // (overlapping intervals: only dnagpu_profile_get's union accounts for them)
}

void gemm_profile_reset(InvWorkspace& ws) {
    ws.prof.flops = 0.0;
    ws.prof.gemm_ms = 0.0;
    ws.prof.launches = 0;
    ws.prof.used = 0;
    ws.prof.side_used = 0;
    ws.prof.open = false;
}

// Opt-in (DNAGPU_LOOKAHEAD=1 / dnagpu_debug_set_lookahead): measured (round 3, profiles/r03_lookahead.txt), bit-identical and no gain --
// 2 458 against 2 460 ms per cfg3 step, 3 079 against 3 077 with one chain and no batches, 402.8 against 402.1 ms for cfg2: a leaf or a
// few-tile product behind a launch that holds every workgroup slot waits for a slot until a tile of that launch ends (0.5 - 1 ms at these
// k), whatever the priorities of the two streams, so the diagonal block advances no faster beside the update than after it.
static std::atomic<int> g_lookahead{[] {
    const char* e = getenv("DNAGPU_LOOKAHEAD");
    return e ? atoi(e) : 0;
}()};
// a trailing update is split only if the part that goes to the side stream has at least this many 128-tiles (of one member)
static std::atomic<long> g_la_min_tiles{getenv("DNAGPU_LOOKAHEAD_MIN_TILES") ? atol(getenv("DNAGPU_LOOKAHEAD_MIN_TILES")) : 1024};
int lookahead_set(int on, long min_tiles) {
    if (min_tiles >= 0) g_la_min_tiles.store(min_tiles);
    return g_lookahead.exchange(on);
}

namespace {

// The recursion over 128-tiles.  F (ld): the matrix, factored in place; X (ldx): L^-1; P (ldp): the L21 panels.
struct Rec {
    InvWorkspace& ws;
    double* F; int ld;
    double* X; int ldx;
    double* P; int ldp;
    bool dry;  // planning pass: only build the tile-order tables, launch nothing
    DagBuilder* rec = nullptr;   // recording pass (tile_dag.h): F / X / P are the builder's symbolic buffers, nothing is launched

    double* f(int rt, int ct) { return F + (size_t)ct * 128 * ld + (size_t)rt * 128; }
    double* x(int rt, int ct) { return X + (size_t)ct * 128 * ldx + (size_t)rt * 128; }
    // (batched calls: P is addressed relative to the diagonal block being factored, see InvBatch)
    int p_o = 0;
    double* w(int rt, int ct) { return P + (size_t)(ct - p_o) * 128 * ldp + (size_t)(rt - p_o) * 128; }
    struct PLocal {     // p_o = o for the lifetime of the object, when the call is batched
        Rec& r; int old;
        PLocal(Rec& r_, int o) : r(r_), old(r_.p_o) { if (r.ws.batch.nb > 1) r.p_o = o; }
        ~PLocal() { r.p_o = old; }
    };

    // ---- look-ahead (sym_inverse.h) ----
    // an operand as an address box: which buffer, which element rows / columns of it
    struct Box {
        int buf = -1;
        long r0 = 0, r1 = 0, c0 = 0, c1 = 0;
        bool meets(const Box& o) const { return buf >= 0 && buf == o.buf && r0 < o.r1 && o.r0 < r1 && c0 < o.c1 && o.c0 < c1; }
    };
    Box box_of(const double* p, long rows, long cols) const {
        // the buffer: the one of F / X / P that starts closest below p (distinct allocations)
        const double* base[3] = {F, X, P};
        const int lds[3] = {ld, ldx, ldp};
        Box b;
        for (int q = 0; q < 3; ++q)
            if (base[q] && p >= base[q] && (b.buf < 0 || base[q] > base[b.buf])) b.buf = q;
        if (b.buf < 0) return b;
        const long off = (long)(p - base[b.buf]);
        b.c0 = off / lds[b.buf];
        b.r0 = off % lds[b.buf];
        b.r1 = b.r0 + rows;
        b.c1 = b.c0 + cols;
        return b;
    }
    struct Boxes { Box A, B, C; };
    // the operands of (tile columns [jlo, jhi) of) a launch; k ranges and the triangle are not looked at: a superset
    Boxes boxes_of(const GemmArgs& a, int akc, int bkc, int jlo, int jhi) const {
        const long i0 = a.lower ? (long)jlo * 128 : 0, rows = (long)a.mt * 128 - i0, j0 = (long)jlo * 128, cols = (long)(jhi - jlo) * 128;
        Boxes b;
        b.C = a.mirror ? box_of(a.C, (long)a.mt * 128, (long)a.nt * 128) : box_of(a.C + (size_t)j0 * a.ldc + i0, rows, cols);
        b.A = akc ? box_of(a.A + (size_t)i0 * a.lda, a.K, rows) : box_of(a.A + i0, rows, a.K);
        b.B = bkc ? box_of(a.B + (size_t)j0 * a.ldb, a.K, cols) : box_of(a.B + j0, cols, a.K);
        return b;
    }
    struct Pending {
        hipEvent_t done;
        int side;
        Boxes b;
    };
    std::vector<Pending> pending;
    static bool conflict(const Boxes& later, const Boxes& earlier) {
        return later.C.meets(earlier.C) || later.A.meets(earlier.C) || later.B.meets(earlier.C) || later.C.meets(earlier.A) || later.C.meets(earlier.B);
    }
    hipEvent_t la_event() {
        if (ws.la_used + 1 > ws.la_events.size()) {
            hipEvent_t ev = nullptr;
            inv_note_error(ws, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "look-ahead event");
            ws.la_events.push_back(ev);
        }
        return ws.la_events[ws.la_used++];
    }
    // the chain's stream waits for the side launches that a launch with these operands must not overtake
    void before_main(const Boxes& b) {
        for (size_t i = 0; i < pending.size();)
            if (conflict(b, pending[i].b)) {
                inv_note_error(ws, hipStreamWaitEvent(ws.stream, pending[i].done, 0), "look-ahead join");
                pending.erase(pending.begin() + i);
            } else {
                ++i;
            }
    }
    void join_all() {
        for (const Pending& pd : pending) inv_note_error(ws, hipStreamWaitEvent(ws.stream, pd.done, 0), "look-ahead join");
        pending.clear();
    }
    bool la_possible() const { return g_lookahead.load() != 0 && !rec && ws.dist_world == 1 && !ws.fuse; }

    void gemm(InvWorkspace& w_, GemmArgs a, int akc, int bkc) {
        if (rec) {
            rec->add_gemm(a, akc, bkc);
        } else if (dry) {
            gemm_attach_order(w_, a);
        } else {
            if (!pending.empty()) before_main(boxes_of(a, akc, bkc, 0, a.nt));
            dnagpu::gemm(w_, a, akc, bkc);
        }
    }

    // A trailing update  C -= W W^T  (r x r lower tiles) after which the chain's next work -- the next diagonal block and its panel --
    // only touches the first `la` tile columns of C: those go out on the chain's stream, the others on a side stream behind the
    // panel that produced W.  Whatever touches them later waits for them (before_main).
    void trailing(GemmArgs a, int la) {
        const int r = a.mt;
        const long rest = (long)(r - la) * (r - la + 1) / 2;
        if (!la_possible() || la <= 0 || la >= r || rest < g_la_min_tiles.load()) {
            gemm(ws, a, 0, 0);
            return;
        }
        if (dry) {
            gemm_attach_order(ws, a, 0, la);
            gemm_attach_order(ws, a, la, r);
            return;
        }
        if (ws.err != hipSuccess) return;
        int side = -1;
        for (int q = 0; q < InvWorkspace::LA_SIDES && side < 0; ++q) {
            bool busy = false;
            for (const Pending& pd : pending) busy = busy || pd.side == q;
            if (!busy) side = q;
        }
        if (side < 0) side = (int)(ws.la_launches % InvWorkspace::LA_SIDES);
        if (!ws.side[side]) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo: the numerically greatest = least urgent
            hipError_t e = hipStreamCreateWithPriority(&ws.side[side], hipStreamNonBlocking, lo);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                ws.side[side] = nullptr;
                gemm(ws, a, 0, 0);      // no stream to be had: the plain launch
                return;
            }
        }
        gemm_flush(ws);
        gemm_profile_close(ws);         // (the event that follows ends a run of GEMM launches on the chain's stream)
        hipStream_t ss = ws.side[side];
        // the side part starts when everything up to here -- the panel W -- is done, and after the side launches it collides with
        hipEvent_t fork = la_event();
        inv_note_error(ws, hipEventRecord(fork, ws.stream), "look-ahead fork");
        inv_note_error(ws, hipStreamWaitEvent(ss, fork, 0), "look-ahead fork");
        const Boxes bb = boxes_of(a, 0, 0, la, r);
        for (const Pending& pd : pending)
            if (pd.side != side && (conflict(bb, pd.b) || conflict(pd.b, bb))) inv_note_error(ws, hipStreamWaitEvent(ss, pd.done, 0), "look-ahead order");
        before_main(boxes_of(a, 0, 0, 0, la));
        gemm_part(ws, a, 0, 0, ws.stream, 0, la, true);
        gemm_part(ws, a, 0, 0, ss, la, r, false);
        Pending pd;
        pd.done = la_event();
        pd.side = side;
        pd.b = bb;
        inv_note_error(ws, hipEventRecord(pd.done, ss), "look-ahead event");
        pending.push_back(pd);
    }

    // W21 = A21 * X11^T (L21 = A21 L11^-T) for the r tile rows below the h x h block at o, then A22 -= W21 * W21^T (lower tiles)
    void eliminate(int o, int h, int r, int la = 0) {
        GemmArgs a;
        a.A = f(o + h, o); a.lda = ld;
        a.B = x(o, o); a.ldb = ldx;
        a.C = w(o + h, o); a.ldc = ldp;
        a.mt = r; a.nt = h; a.K = h * 128;
        a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_LE_J; a.lower = 0; a.mirror = 0;
        gemm(ws, a, 0, 0);
        a.A = w(o + h, o); a.lda = ldp;
        a.B = w(o + h, o); a.ldb = ldp;
        a.C = f(o + h, o + h); a.ldc = ld;
        a.mt = r; a.nt = r; a.K = h * 128;
        a.alpha = -1.0; a.beta = 1.0; a.kmode = KM_FULL; a.lower = 1; a.mirror = 0;
        trailing(a, la);
    }

    // Cholesky factor and its inverse of the s x s tile block at o: F keeps T21 = L21 L11^-1 below the diagonal, X = L^-1
    void node(int o, int s) {
        if (s == 1) {
            if (rec) {
                rec->add_leaf(f(o, o), x(o, o), o);
            } else if (!dry && ws.err == hipSuccess) {
                gemm_profile_close(ws);
                if (!pending.empty()) {
                    Boxes b;
                    b.A = box_of(f(o, o), 128, 128);
                    b.C = box_of(x(o, o), 128, 128);
                    before_main(b);
                }
                if (ws.batch.nb > 1) {
                    LeafBatch lb;
                    lb.nb = ws.batch.nb;
                    if (!batch_offsets(ws.batch, F, lb.dA) || !batch_offsets(ws.batch, X, lb.dX)) {
                        inv_note_error(ws, hipErrorInvalidValue, "batched leaf: a matrix outside the registered buffers");
                        return;
                    }
                    launch_leaf(F, ld, X, ldx, o * 128, ws.info, ws.stream, &lb);
                } else {
                    launch_leaf(F, ld, X, ldx, o * 128, ws.info, ws.stream);
                }
                inv_note_error(ws, hipGetLastError(), "leaf launch");
            }
            return;
        }
        int h = s / 2;
        int r = s - h;
        node(o, h);
        eliminate(o, h, r, r / 2);      // (what follows -- the left half of the right block and its panel -- stays in its first r / 2 columns)
        node(o + h, r);
        GemmArgs a;
        // T21 = W21 * X11  -> stored where A21 was
        a.A = w(o + h, o); a.lda = ldp;
        a.B = x(o, o); a.ldb = ldx;
        a.C = f(o + h, o); a.ldc = ld;
        a.mt = r; a.nt = h; a.K = h * 128;
        a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_J; a.lower = 0; a.mirror = 0;
        gemm(ws, a, 0, 1);
        // X21 = -X22 * T21
        a.A = x(o + h, o + h); a.lda = ldx;
        a.B = f(o + h, o); a.ldb = ld;
        a.C = x(o + h, o); a.ldc = ldx;
        a.mt = r; a.nt = h; a.K = r * 128;
        a.alpha = -1.0; a.beta = 0.0; a.kmode = KM_LE_I; a.lower = 0; a.mirror = 0;
        gemm(ws, a, 0, 1);
    }

    // Eliminates the first ti tile rows / columns; the trailing tj x tj tiles receive every update and are never factored:
    // they end as the Schur complement.  Each step factors (and inverts, for the panel solve) a leading h x h block of what
    // is left; potrf + trtri of the block is 2/3 h^3 against h^2 r + h r^2 for the panel, so the blocks stay a fraction
    // `split` of the remainder: total 0.381 / 0.351 / 0.342 / 0.338 n^3 at split 1/2, 1/3, 1/4, 1/5 (a Cholesky factorisation
    // alone is n^3/3).  Measured on cfg3 (n = 20 k): 7.18 / 6.97 / 6.89 / 6.82 / 6.79 s per step at 0.5 / 0.33 / 0.25 / 0.2 / 0.15
    // (smaller blocks = more, smaller launches): 0.2, override DNAGPU_SCHUR_SPLIT.
    static int spine_step(int si, double split) { return si <= 12 ? si : std::max(1, std::min(si, (int)(si * split + 0.5))); }
    void schur(int ti, int tj, double split) {
        int o = 0, si = ti;
        while (si > 0) {
            int h = spine_step(si, split);
            node(o, h);
            int r = si - h + tj;
            if (r > 0) eliminate(o, h, r, si - h > 0 ? spine_step(si - h, split) : 0);
            o += h;
            si -= h;
        }
    }

    // The same elimination with its factor KEPT in X (ldx) as one block lower triangular matrix: the diagonal blocks hold the INVERSES
    // of the factor's diagonal blocks (what node() leaves in X anyway), the blocks below them the factor's panels L_(below, b) -- at the
    // very place where the corresponding block of L^-1 goes once spine_finish() runs.  (The panels of the levels INSIDE a diagonal
    // block still go to P: they are dead when node() returns.)
    void spine(int ti, int tj, double split) {
        int o = 0, si = ti;
        while (si > 0) {
            int h = spine_step(si, split);
            {
                PLocal pl(*this, o);
                node(o, h);
            }
            int r = si - h + tj;
            if (r > 0) {
                GemmArgs a;
                a.A = f(o + h, o); a.lda = ld;
                a.B = x(o, o); a.ldb = ldx;
                a.C = x(o + h, o); a.ldc = ldx;
                a.mt = r; a.nt = h; a.K = h * 128;
                a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_LE_J; a.lower = 0; a.mirror = 0;
                gemm(ws, a, 0, 0);
                a.A = x(o + h, o); a.lda = ldx;
                a.B = x(o + h, o); a.ldb = ldx;
                a.C = f(o + h, o + h); a.ldc = ld;
                a.mt = r; a.nt = r; a.K = h * 128;
                a.alpha = -1.0; a.beta = 1.0; a.kmode = KM_FULL; a.lower = 1; a.mirror = 0;
                trailing(a, si - h > 0 ? spine_step(si - h, split) : 0);
            }
            o += h;
            si -= h;
        }
    }
    // ... and from that to L^-1 of the whole (ti + tj) x (ti + tj) matrix, the trailing tj x tj block of X holding the inverse of ITS
    // factor already: block column by block column from the last to the first,  T = L_(below, b) X_bb,  X_(below, b) = -X_(below, below) T
    // -- the two products node() issues when it returns from its right child, n^3 / 3 flops in all.
    void spine_finish(int ti, int tj, double split) {
        std::vector<std::pair<int, int>> blocks;
        for (int o = 0, si = ti; si > 0;) {
            int h = spine_step(si, split);
            blocks.emplace_back(o, h);
            o += h;
            si -= h;
        }
        const int T = ti + tj;
        for (size_t q = blocks.size(); q-- > 0;) {
            const int o = blocks[q].first, h = blocks[q].second, r = T - (o + h);
            if (r <= 0) continue;
            GemmArgs a;
            a.A = x(o + h, o); a.lda = ldx;
            a.B = x(o, o); a.ldb = ldx;
            a.C = f(o + h, o); a.ldc = ld;
            a.mt = r; a.nt = h; a.K = h * 128;
            a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_J; a.lower = 0; a.mirror = 0;
            gemm(ws, a, 0, 1);
            a.A = x(o + h, o + h); a.lda = ldx;
            a.B = f(o + h, o); a.ldb = ld;
            a.C = x(o + h, o); a.ldc = ldx;
            a.mt = r; a.nt = h; a.K = r * 128;
            a.alpha = -1.0; a.beta = 0.0; a.kmode = KM_LE_I; a.lower = 0; a.mirror = 0;
            gemm(ws, a, 0, 1);
        }
    }
};

}  // namespace

// ---- the tile-DAG path (tile_dag.h): the op sequence of a driver below recorded once per shape, then one launch per call ----
namespace {

struct DagCache {
    std::mutex m;
    std::map<std::tuple<int, int, int, int, int, int, int, int, int, int, int, long>, std::shared_ptr<DagGraph>> graphs;
};
DagCache& dag_cache() {
    static DagCache* c = new DagCache();     // (never destroyed: device memory must not be freed after the runtime has gone)
    return *c;
}

// Opt-in (DNAGPU_DAG=1 / dnagpu_debug_set_tile_dag): measured (round 3, profiles/r03_dag_schedulers.txt), the DAG path reproduces the
// per-product path bit for bit and does not beat it -- one chain 3.12 s against 3.07 s per cfg3 step, four chains 2.64 - 2.75 s against
// 2.55 s: the factorisation is bound by its critical path, and workers that wait on it hold slots the other chains have work for.
std::atomic<int> g_dag_mode{[] {
    const char* e = getenv("DNAGPU_DAG");
    return e ? atoi(e) : 0;
}()};
std::atomic<int> g_dag_min_tiles{[] {
    const char* e = getenv("DNAGPU_DAG_MIN_TILES");
    return e ? atoi(e) : 2;
}()};
// 0 = recorded order, 1 = list-scheduling order (default); workers: workgroups per launch unless the chain says otherwise (InvWorkspace::dag_workers)
const int g_dag_reorder = getenv("DNAGPU_DAG_ORDER") ? atoi(getenv("DNAGPU_DAG_ORDER")) : 1;
const int g_dag_workers = getenv("DNAGPU_DAG_WORKERS") ? atoi(getenv("DNAGPU_DAG_WORKERS")) : 512;

std::atomic<int> g_dag_trace_serial{0};

enum DagKind { DK_INVERSE = 1, DK_SCHUR = 2, DK_SCHUR_KEEP = 3, DK_COMPLETE = 4, DK_SPINE = 5, DK_SPINE_KEPT = 6, DK_SPINE_FINISH = 7 };

template <class Ops>
std::shared_ptr<DagGraph> dag_record(InvWorkspace& ws, int ld, int ldx, int ldp, int ldwk, Ops&& ops) {
    const int lds[DAG_MAX_BUFS] = {ld, ldx, ldp, ldwk > 0 ? ldwk : 128};
    DagBuilder b(DAG_MAX_BUFS, lds, g_small_tiles.load());
    Rec rec{ws, b.base(0), ld, b.base(1), ldx, b.base(2), ldp, false, &b};
    ops(rec, (const double*)b.base(3));
    return b.finish(g_dag_reorder, g_dag_workers);
}

// true: the call went out (or was dropped after a latched error) as one DAG launch; false: the caller launches product by product
template <class Ops>
bool run_dag(InvWorkspace& ws, int kind, int ti, int tj, int what, double* F, int ld, double* X, int ldx, double* P, int ldp, const double* WK,
             int ldwk, Ops&& ops) {
    if (!ws.dag || !g_dag_mode.load() || ws.dist_world > 1 || ws.batch.nb > 1 || ti + tj < g_dag_min_tiles.load()) return false;
    if (ws.err != hipSuccess) return true;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_tuple(dev, kind, ti, tj, what, ld, ldx, ldp, ldwk, (int)(schur_split() * 1000.0 + 0.5), g_dag_reorder, g_small_tiles.load());
    std::shared_ptr<DagGraph> g;
    {
        DagCache& c = dag_cache();
        std::lock_guard<std::mutex> lock(c.m);
        auto it = c.graphs.find(key);
        if (it == c.graphs.end()) {
            g = dag_record(ws, ld, ldx, ldp, ldwk, ops);
            hipError_t e = dag_upload(*g);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                inv_note_error(ws, e, "tile DAG upload");
                return true;
            }
            static const bool verbose = getenv("DNAGPU_DAG_VERBOSE") != nullptr;
            if (verbose)
                fprintf(stderr, "dnagpu: tile DAG kind %d ti %d tj %d: %zu tasks (%u products, %u leaves), %zu dependency runs, work %.1f ms / 512 = %.2f ms, "
                        "critical path %.2f ms, simulated %.2f ms\n", kind, ti, tj, g->tasks.size(), g->n_products, g->n_leaves, g->deps.size(),
                        g->sim_work_us / 1e3, g->sim_work_us / 512e3, g->critical_path_us / 1e3, g->sim_makespan_us / 1e3);
            c.graphs.emplace(key, g);
        } else {
            g = it->second;
        }
    }
    if (g->tasks.empty()) return true;
    // (error-path tests, dnagpu_debug_fail_allocation: the per-product path allocates a tile-order table per launch shape; this path
    //  allocates per graph and per workspace only, so the countdown also runs over its launches)
    if (g_fault_countdown.load() > 0 && g_fault_countdown.fetch_sub(1) == 1) {
        inv_note_error(ws, hipErrorOutOfMemory, "tile DAG allocation");
        return true;
    }
    // completion flags (raised by writing the launch's epoch: never cleared) and the ticket counter of this chain.  Zeroed on the
    // chain's own stream: a plain hipMemset runs on the legacy stream, which a non-blocking stream does not wait for -- the first
    // launch would take tickets from a counter that is zeroed underneath it (seen: 657 of 5 260 tasks never ran)
    if (ws.dag_flags_cap < g->nids) {
        hipStreamSynchronize(ws.stream);
        if (ws.dag_flags) hipFree(ws.dag_flags);
        ws.dag_flags = nullptr;
        ws.dag_flags_cap = 0;
        const size_t want = std::max<size_t>((size_t)g->nids * 2, (size_t)1 << 20);
        hipError_t e = dnagpu::poison_malloc(&ws.dag_flags, want * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(ws.dag_flags, 0, want * sizeof(uint32_t), ws.stream);
        if (e == hipSuccess && !ws.dag_ticket) {
            e = dnagpu::poison_malloc(&ws.dag_ticket, sizeof(unsigned long long));
            if (e == hipSuccess) e = hipMemsetAsync(ws.dag_ticket, 0, sizeof(unsigned long long), ws.stream);
            ws.dag_ticket_base = 0;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            inv_note_error(ws, e, "tile DAG flags allocation");
            return true;
        }
        ws.dag_flags_cap = want;
        ws.dag_epoch = 0;
    }
    gemm_flush(ws);
    GemmProfile& p = ws.prof;
    if (p.enabled) {
        if (!p.open) {
            profile_event(p, ws.stream);
            p.open = true;
        }
        p.flops += g->flops;
        p.launches++;
    }
    DagLaunch L;
    L.tasks = g->d_tasks;
    L.deps = g->d_deps;
    L.ntasks = (uint32_t)g->tasks.size();
    L.epoch = ++ws.dag_epoch;
    L.flags = ws.dag_flags;
    L.ticket = ws.dag_ticket;
    L.ticket_base = ws.dag_ticket_base;
    static const int paranoid = getenv("DNAGPU_DAG_PARANOID") ? atoi(getenv("DNAGPU_DAG_PARANOID")) : 0;
    L.paranoid = paranoid;
    L.workers = (int)std::min<size_t>(g->tasks.size(), (size_t)std::max(1, ws.dag_workers > 0 ? ws.dag_workers : g_dag_workers));
    L.buf0 = F; L.buf1 = X; L.buf2 = P; L.buf3 = const_cast<double*>(WK);
    L.ld0 = ld; L.ld1 = ldx; L.ld2 = ldp; L.ld3 = ldwk > 0 ? ldwk : 128;
    L.info = ws.info;
    // DNAGPU_DAG_TRACE=<prefix>: per-task clocks of every launch, written to <prefix>.<launch>.bin (tools/dag_trace_report.py); diagnostic, synchronous
    static const char* trace_prefix = getenv("DNAGPU_DAG_TRACE");
    L.trace = nullptr;
    if (trace_prefix && dnagpu::poison_malloc(&L.trace, (size_t)L.ntasks * 4 * sizeof(unsigned long long)) == hipSuccess)
        hipMemsetAsync(L.trace, 0, (size_t)L.ntasks * 4 * sizeof(unsigned long long), ws.stream);
    launch_tile_dag(L, ws.stream);
    inv_note_error(ws, hipGetLastError(), "tile DAG launch");
    if (L.trace) {
        hipStreamSynchronize(ws.stream);
        std::vector<unsigned long long> tr((size_t)L.ntasks * 4);
        hipMemcpy(tr.data(), L.trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipFree(L.trace);
        char name[512];
        snprintf(name, sizeof(name), "%s.%04d.bin", trace_prefix, g_dag_trace_serial.fetch_add(1));
        if (FILE* f = fopen(name, "wb")) {
            const uint64_t hdr[8] = {L.ntasks, (uint64_t)kind, (uint64_t)ti, (uint64_t)tj, (uint64_t)what, sizeof(DagTask), 0, 0};
            fwrite(hdr, sizeof(hdr), 1, f);
            fwrite(tr.data(), sizeof(unsigned long long), tr.size(), f);
            fwrite(g->tasks.data(), sizeof(DagTask), g->tasks.size(), f);
            fclose(f);
        }
    }
    ws.dag_ticket_base += (unsigned long long)L.ntasks + (unsigned long long)L.workers;     // (every worker takes one ticket too many)
    ws.dag_launches++;
    ws.dag_tasks += L.ntasks;
    return true;
}

// key of a driver call's shape in InvWorkspace::planned: fields wide enough for any matrix this library can hold (2^24 tiles a side)
inline uint64_t plan_key(int family, int what, int ti, int tj) {
    return ((uint64_t)family << 56) | ((uint64_t)(what & 0xf) << 52) | ((uint64_t)(uint32_t)ti << 26) | (uint64_t)(uint32_t)tj;
}

// the per-product path: planning pass (tile-order tables) on first use of the shape, then the launches
template <class Ops>
void run_products(InvWorkspace& ws, uint64_t key, double* F, int ld, double* X, int ldx, double* P, int ldp, const double* WK, Ops&& ops) {
    for (int pass = ws.planned.count(key) ? 1 : 0; pass < 2; ++pass) {
        Rec rec{ws, F, ld, X, ldx, P, ldp, pass == 0};
        ws.la_used = 0;
        ops(rec, WK);
        rec.join_all();         // (look-ahead: everything this call put on the side streams is part of it)
        if (ws.err != hipSuccess) return;
    }
    ws.planned.insert(key);
}

}  // namespace

int dag_mode_set(int on) { return g_dag_mode.exchange(on); }

void sym_inverse_async(InvWorkspace& ws, double* F, uint32_t n, uint32_t np, bool scale_to_unity, bool reset_info) {
    if (reset_info) inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");  // 0x7f7f7f7f = "no failure" sentinel for atomicMin
    if (scale_to_unity) {
        launch_diag_rsqrt(F, ws.svec, n, np, ws.stream);
        launch_scale_sym(F, ws.svec, n, np, 1, ws.stream);
    }
    const int T = (int)(np / 128);
    // DNAGPU_POISON_ALLOC=2 (diagnostic): the scratch of an inverse (X: L^-1, W: the panels) starts as NaN -- a tile that is read before
    // this call wrote it shows up in the result instead of passing as whatever the previous call on this chain left there
    static const bool poison_scratch = getenv("DNAGPU_POISON_ALLOC") && atoi(getenv("DNAGPU_POISON_ALLOC")) >= 2;
    if (poison_scratch) {
        inv_note_error(ws, hipMemsetAsync(ws.X, 0xFF, (size_t)np * np * sizeof(double), ws.stream), "poison");
        inv_note_error(ws, hipMemsetAsync(ws.W, 0xFF, (size_t)np * np * sizeof(double), ws.stream), "poison");
    }
    // potrf + trtri by the recursion, then Ninv = X^T X (lauum), both triangles
    auto ops = [&](Rec& rec, const double*) {
        rec.node(0, T);
        GemmArgs a;
        a.A = rec.X; a.lda = rec.ldx;
        a.B = rec.X; a.ldb = rec.ldx;
        a.C = rec.F; a.ldc = rec.ld;
        a.mt = T; a.nt = T; a.K = T * 128;
        a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_I; a.lower = 1; a.mirror = 1;
        rec.gemm(ws, a, 1, 1);
    };
    if (!run_dag(ws, DK_INVERSE, T, 0, 0, F, (int)np, ws.X, (int)np, ws.W, (int)np, nullptr, 0, ops))
        run_products(ws, plan_key(DK_INVERSE, 0, T, 0), F, (int)np, ws.X, (int)np, ws.W, (int)np, nullptr, ops);     // (tables first: the blocking uploads never sit between kernels)
    gemm_profile_close(ws);
    if (scale_to_unity) launch_scale_sym(F, ws.svec, n, np, 0, ws.stream);
    inv_note_error(ws, hipGetLastError(), "inverse: scaling launch");
    inv_note_error(ws, hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, ws.stream), "info copy");
}

double schur_split() {
    static double v = [] {
        const char* e = getenv("DNAGPU_SCHUR_SPLIT");
        double x = e ? atof(e) : 0.0;
        return (x > 0.0 && x <= 1.0) ? x : 0.2;
    }();
    return v;
}

void sym_schur_keep_async(InvWorkspace& ws, double* F, double* X, int ld, int ti, int tj) {
    inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");
    auto ops = [&](Rec& rec, const double*) {
        if (ti > 0) rec.node(0, ti);
        if (ti > 0 && tj > 0) rec.eliminate(0, ti, tj);
    };
    if (!run_dag(ws, DK_SCHUR_KEEP, ti, tj, 0, F, ld, X, ld, ws.W, ld, nullptr, 0, ops))
        run_products(ws, plan_key(DK_SCHUR_KEEP, 0, ti, tj), F, ld, X, ld, ws.W, ld, nullptr, ops);
    gemm_profile_close(ws);
}

void sym_complete_async(InvWorkspace& ws, double* F, double* X, int ld, const double* WK, int ldwk, int ti, int tj, int what) {
    if (what & 1) inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");
    const int T = ti + tj;
    auto ops = [&](Rec& rec, const double* wk) {
        GemmArgs a;
        if (what & 1) {
            rec.node(ti, tj);
            if (ti > 0) {
                // T_KI = L_KI * X_II -> the kept rows of F;  X_KI = -X_KK * T_KI
                a.A = wk; a.lda = ldwk;
                a.B = rec.x(0, 0); a.ldb = rec.ldx;
                a.C = rec.f(ti, 0); a.ldc = rec.ld;
                a.mt = tj; a.nt = ti; a.K = ti * 128;
                a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_J; a.lower = 0; a.mirror = 0;
                rec.gemm(ws, a, 0, 1);
                a.A = rec.x(ti, ti); a.lda = rec.ldx;
                a.B = rec.f(ti, 0); a.ldb = rec.ld;
                a.C = rec.x(ti, 0); a.ldc = rec.ldx;
                a.mt = tj; a.nt = ti; a.K = tj * 128;
                a.alpha = -1.0; a.beta = 0.0; a.kmode = KM_LE_I; a.lower = 0; a.mirror = 0;
                rec.gemm(ws, a, 0, 1);
            }
        }
        if (what & 2) {
            // inverse = X^T X, both triangles
            a.A = rec.X; a.lda = rec.ldx;
            a.B = rec.X; a.ldb = rec.ldx;
            a.C = rec.F; a.ldc = rec.ld;
            a.mt = T; a.nt = T; a.K = T * 128;
            a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_I; a.lower = 1; a.mirror = 1;
            rec.gemm(ws, a, 1, 1);
        }
    };
    if (!run_dag(ws, DK_COMPLETE, ti, tj, what & 3, F, ld, X, ld, ws.W, ld, WK, ldwk, ops))
        run_products(ws, plan_key(DK_COMPLETE, what & 3, ti, tj), F, ld, X, ld, ws.W, ld, WK, ops);
    gemm_profile_close(ws);
}

void sym_spine_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj) {
    inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");
    const double split = schur_split();
    auto ops = [&](Rec& rec, const double*) { rec.spine(ti, tj, split); };
    if (!run_dag(ws, DK_SPINE, ti, tj, 0, F, ld, S, ld, ws.W, ld, nullptr, 0, ops))
        run_products(ws, plan_key(DK_SPINE, 0, ti, tj), F, ld, S, ld, ws.W, ld, nullptr, ops);
    gemm_profile_close(ws);
}

void sym_spine_kept_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj) {
    inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");
    auto ops = [&](Rec& rec, const double*) {
        Rec::PLocal pl(rec, ti);
        rec.node(ti, tj);
    };
    if (!run_dag(ws, DK_SPINE_KEPT, ti, tj, 0, F, ld, S, ld, ws.W, ld, nullptr, 0, ops))
        run_products(ws, plan_key(DK_SPINE_KEPT, 0, ti, tj), F, ld, S, ld, ws.W, ld, nullptr, ops);
    gemm_profile_close(ws);
}

void sym_spine_finish_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj) {
    const double split = schur_split();
    const int T = ti + tj;
    auto ops = [&](Rec& rec, const double*) {
        rec.spine_finish(ti, tj, split);
        GemmArgs a;
        a.A = rec.X; a.lda = rec.ldx;
        a.B = rec.X; a.ldb = rec.ldx;
        a.C = rec.F; a.ldc = rec.ld;
        a.mt = T; a.nt = T; a.K = T * 128;
        a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_I; a.lower = 1; a.mirror = 1;
        rec.gemm(ws, a, 1, 1);
    };
    if (!run_dag(ws, DK_SPINE_FINISH, ti, tj, 0, F, ld, S, ld, ws.W, ld, nullptr, 0, ops))
        run_products(ws, plan_key(DK_SPINE_FINISH, 0, ti, tj), F, ld, S, ld, ws.W, ld, nullptr, ops);
    gemm_profile_close(ws);
}

std::vector<std::pair<int, int>> sym_spine_blocks(int ti) {
    std::vector<std::pair<int, int>> blocks;
    const double split = schur_split();
    for (int o = 0, si = ti; si > 0;) {
        int h = si <= 12 ? si : std::max(1, std::min(si, (int)(si * split + 0.5)));
        blocks.emplace_back(o, h);
        o += h;
        si -= h;
    }
    return blocks;
}

void sym_schur_async(InvWorkspace& ws, double* F, int ld, double* P, int ldp, int ti, int tj) {
    inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");
    const int ldx = ti * 128;
    const double split = schur_split();
    auto ops = [&](Rec& rec, const double*) { rec.schur(ti, tj, split); };
    if (!run_dag(ws, DK_SCHUR, ti, tj, 0, F, ld, ws.X, ldx, P, ldp, nullptr, 0, ops))
        run_products(ws, plan_key(DK_SCHUR, 0, ti, tj), F, ld, ws.X, ldx, P, ldp, nullptr, ops);
    gemm_profile_close(ws);
}

// CPU self-test of the dependency analysis (no device): the recorded sequence of `kind` for ti + tj tiles is run on host buffers
// in its recorded order, and then -- from the same inputs -- in the launch order, a random admissible order and the most
// out-of-order one the completion flags admit.  Returns the number of orders whose results differ from the recorded order's in
// any bit (0 = the flags carry every dependency), -1 if an order stalls.
int dag_selftest(int kind, int ti, int tj, int what, uint64_t seed, double* stats) {
    InvWorkspace ws;      // (never touched by a recording pass)
    const int T = ti + tj, np = T * 128;
    const int ld = np, ldx = kind == DK_SCHUR ? std::max(1, ti) * 128 : np, ldp = np, ldwk = std::max(1, tj) * 128;
    const double split = schur_split();
    auto ops = [&](Rec& rec, const double* wk) {
        GemmArgs a;
        switch (kind) {
            case DK_INVERSE:
                rec.node(0, T);
                break;
            case DK_SCHUR: rec.schur(ti, tj, split); return;
            case DK_SCHUR_KEEP:
                if (ti > 0) rec.node(0, ti);
                if (ti > 0 && tj > 0) rec.eliminate(0, ti, tj);
                return;
            case DK_SPINE: rec.spine(ti, tj, split); return;
            case DK_SPINE_KEPT: rec.node(ti, tj); return;
            case DK_SPINE_FINISH: rec.spine_finish(ti, tj, split); break;
            case DK_COMPLETE:
                if (what & 1) {
                    rec.node(ti, tj);
                    if (ti > 0) {
                        a.A = wk; a.lda = ldwk;
                        a.B = rec.x(0, 0); a.ldb = rec.ldx;
                        a.C = rec.f(ti, 0); a.ldc = rec.ld;
                        a.mt = tj; a.nt = ti; a.K = ti * 128;
                        a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_J; a.lower = 0; a.mirror = 0;
                        rec.gemm(ws, a, 0, 1);
                        a.A = rec.x(ti, ti); a.lda = rec.ldx;
                        a.B = rec.f(ti, 0); a.ldb = rec.ld;
                        a.C = rec.x(ti, 0); a.ldc = rec.ldx;
                        a.mt = tj; a.nt = ti; a.K = tj * 128;
                        a.alpha = -1.0; a.beta = 0.0; a.kmode = KM_LE_I; a.lower = 0; a.mirror = 0;
                        rec.gemm(ws, a, 0, 1);
                    }
                }
                if (!(what & 2)) return;
                break;
            default: return;
        }
        a.A = rec.X; a.lda = rec.ldx;
        a.B = rec.X; a.ldb = rec.ldx;
        a.C = rec.F; a.ldc = rec.ld;
        a.mt = T; a.nt = T; a.K = T * 128;
        a.alpha = 1.0; a.beta = 0.0; a.kmode = KM_GE_I; a.lower = 1; a.mirror = 1;
        rec.gemm(ws, a, 1, 1);
    };
    const int lds[DAG_MAX_BUFS] = {ld, ldx, ldp, ldwk};
    std::shared_ptr<DagGraph> recorded, scheduled;
    {
        DagBuilder b(DAG_MAX_BUFS, lds, g_small_tiles.load());
        Rec rec{ws, b.base(0), ld, b.base(1), ldx, b.base(2), ldp, false, &b};
        ops(rec, (const double*)b.base(3));
        recorded = b.finish(0, 1);
    }
    {
        DagBuilder b(DAG_MAX_BUFS, lds, g_small_tiles.load());
        Rec rec{ws, b.base(0), ld, b.base(1), ldx, b.base(2), ldp, false, &b};
        ops(rec, (const double*)b.base(3));
        scheduled = b.finish(1, 16);
    }
    if (stats) {
        stats[0] = (double)recorded->tasks.size();
        stats[1] = (double)recorded->deps.size();
        stats[2] = recorded->flops;
        stats[3] = scheduled->sim_makespan_us;
        stats[4] = scheduled->critical_path_us;
        stats[5] = scheduled->sim_work_us;
    }
    // inputs: a diagonally dominant symmetric matrix (both triangles: the sequences read what they were given), benign X / P / WK
    const size_t sz[DAG_MAX_BUFS] = {(size_t)ld * np, (size_t)ldx * np, (size_t)ldp * np, (size_t)ldwk * np};
    std::vector<double> init[DAG_MAX_BUFS];
    uint64_t r = seed * 2862933555777941757ull + 3037000493ull;
    auto rnd = [&] {
        r = r * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    };
    for (int q = 0; q < DAG_MAX_BUFS; ++q) {
        init[q].resize(sz[q]);
        for (double& v : init[q]) v = 0.01 * rnd();
    }
    for (int j = 0; j < np; ++j) {
        for (int i = j + 1; i < np; ++i) init[0][(size_t)i * ld + j] = init[0][(size_t)j * ld + i];
        init[0][(size_t)j * ld + j] = 4.0 + rnd();
    }
    if (kind != DK_INVERSE && kind != DK_SCHUR && kind != DK_SCHUR_KEEP && kind != DK_SPINE)
        for (int j = 0; j < std::min(np, ldx); ++j) init[1][(size_t)j * ldx + j] = 1.0 + 0.1 * rnd();     // (X: a plausible triangular factor inverse)
    auto run = [&](const DagGraph& g, int order, std::vector<double> (&out)[DAG_MAX_BUFS]) {
        double* bp[DAG_MAX_BUFS];
        for (int q = 0; q < DAG_MAX_BUFS; ++q) {
            out[q] = init[q];
            bp[q] = out[q].data();
        }
        return dag_execute_host(g, bp, lds, order, seed + (uint64_t)order);
    };
    std::vector<double> ref[DAG_MAX_BUFS], got[DAG_MAX_BUFS];
    if (!run(*recorded, 0, ref)) return -1;
    int differing = 0;
    const DagGraph* graphs[4] = {scheduled.get(), scheduled.get(), scheduled.get(), recorded.get()};
    const int orders[4] = {0, 1, 2, 2};
    for (int v = 0; v < 4; ++v) {
        if (!run(*graphs[v], orders[v], got)) return -1;
        bool same = true;
        for (int q = 0; q < DAG_MAX_BUFS; ++q) same = same && !memcmp(ref[q].data(), got[q].data(), sz[q] * sizeof(double));
        if (!same) ++differing;
    }
    return differing;
}

}  // namespace dnagpu
