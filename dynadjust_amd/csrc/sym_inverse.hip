#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <cstring>
#include <cstdio>
#include "sym_inverse.h"
#include "la_kernels.h"
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
    if ((e = hipHostMalloc(&ws.info_host, BATCH_MAX * sizeof(int))) != hipSuccess) return e;
    for (int b = 0; b < BATCH_MAX; ++b) ws.info_host[b] = 0;
    return hipSuccess;
}

void inv_workspace_free(InvWorkspace& ws) {
    if (ws.X) hipFree(ws.X);
    if (ws.W) hipFree(ws.W);
    for (int b = 0; b < BATCH_MAX; ++b) {
        if (ws.bX[b]) hipFree(ws.bX[b]);
        if (ws.bW[b]) hipFree(ws.bW[b]);
    }
    if (ws.svec) hipFree(ws.svec);
    if (ws.info) hipFree(ws.info);
    if (ws.dist_stage) hipFree(ws.dist_stage);
    if (ws.info_host) hipHostFree(ws.info_host);
    for (hipEvent_t ev : ws.prof.pool) hipEventDestroy(ev);
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

// 0x7f7f7f7f = "no failure", the sentinel atomicMin works against; skipped while a caller holds the verdict over a run of calls
static void info_reset(InvWorkspace& ws) {
    if (ws.hold_info) return;
    inv_note_error(ws, hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ws.stream), "info reset");
}

hipError_t inv_take_error(InvWorkspace& ws, const char** where) {
    hipError_t e = ws.err;
    if (where) *where = ws.err_where;
    ws.err = hipSuccess;
    ws.err_where = nullptr;
    return e;
}

// fault injection for the error-path tests: countdown to a failing table allocation
static std::atomic<long> g_fault_countdown{0L};
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

// launches with fewer 128-tiles than this run on 64 x 64 block tiles (dnagpu_debug_set_small_tiles: 0 sends
// every launch through the 128-tile throughput kernel -- how the tests compare THAT kernel with the oracle at small orders)
static std::atomic<long> g_small_tiles{(long)SMALL_LAUNCH_TILES};
static std::atomic<long> g_tiny_tiles{(long)TINY_LAUNCH_TILES};
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
    a.tile = total < g_small_tiles.load() ? (total < g_tiny_tiles.load() ? 32 : 64) : 128;
    const uint64_t shape = (uint64_t)a.mt | ((uint64_t)a.nt << 16) | ((uint64_t)a.kmode << 32) | ((uint64_t)(a.lower ? 1 : 0) << 36) |
                           ((uint64_t)(a.tile == 64 ? 1 : 0) << 37) | ((uint64_t)(a.tile == 32 ? 1 : 0) << 39) | ((uint64_t)(a.K / 16) << 40);
    const auto key = std::make_pair(shape, jt_lo < 0 ? 0xffffffffu : ((uint32_t)jt_lo << 16) | (uint32_t)jt_hi);
    auto it = ws.order_cache.find(key);
    if (it == ws.order_cache.end()) {
        std::vector<uint32_t> tab = build_tile_order(a.mt, a.nt, a.K, a.kmode, a.lower, a.tile, jt_lo, jt_hi);
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
        it = ws.order_cache.emplace(key, std::make_pair(dev, (int)tab.size())).first;
    }
    a.order = it->second.first;
    a.grid = it->second.second;
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

void gemm_profile_close(InvWorkspace& ws) {
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
}

void gemm_profile_reset(InvWorkspace& ws) {
    ws.prof.flops = 0.0;
    ws.prof.gemm_ms = 0.0;
    ws.prof.launches = 0;
    ws.prof.used = 0;
    ws.prof.open = false;
}

namespace {

// The recursion over 128-tiles.  F (ld): the matrix, factored in place; X (ldx): L^-1; P (ldp): the L21 panels.
struct Rec {
    InvWorkspace& ws;
    double* F; int ld;
    double* X; int ldx;
    double* P; int ldp;
    bool dry;  // planning pass: only build the tile-order tables, launch nothing

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

    void gemm(InvWorkspace& w_, GemmArgs a, int akc, int bkc) {
        if (dry)
            gemm_attach_order(w_, a);
        else
            dnagpu::gemm(w_, a, akc, bkc);
    }

    // W21 = A21 * X11^T (L21 = A21 L11^-T) for the r tile rows below the h x h block at o, then A22 -= W21 * W21^T (lower tiles)
    void eliminate(int o, int h, int r) {
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
        gemm(ws, a, 0, 0);
    }

    // Cholesky factor and its inverse of the s x s tile block at o: F keeps T21 = L21 L11^-1 below the diagonal, X = L^-1
    void node(int o, int s) {
        if (s == 1) {
            if (!dry && ws.err == hipSuccess) {
                gemm_profile_close(ws);
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
        eliminate(o, h, r);
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
    // (smaller blocks = more, smaller launches): 0.2.
    static int spine_step(int si, double split) { return si <= 12 ? si : std::max(1, std::min(si, (int)(si * split + 0.5))); }
    void schur(int ti, int tj, double split) {
        int o = 0, si = ti;
        while (si > 0) {
            int h = spine_step(si, split);
            node(o, h);
            int r = si - h + tj;
            if (r > 0) eliminate(o, h, r);
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
                gemm(ws, a, 0, 0);
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

namespace {

enum DriverKind { DK_INVERSE = 1, DK_SCHUR = 2, DK_SCHUR_KEEP = 3, DK_COMPLETE = 4, DK_SPINE = 5, DK_SPINE_KEPT = 6, DK_SPINE_FINISH = 7 };

// key of a driver call's shape in InvWorkspace::planned: fields wide enough for any matrix this library can hold (2^24 tiles a side)
inline uint64_t plan_key(int family, int what, int ti, int tj) {
    return ((uint64_t)family << 56) | ((uint64_t)(what & 0xf) << 52) | ((uint64_t)(uint32_t)ti << 26) | (uint64_t)(uint32_t)tj;
}

// the per-product path: planning pass (tile-order tables) on first use of the shape, then the launches
template <class Ops>
void run_products(InvWorkspace& ws, uint64_t key, double* F, int ld, double* X, int ldx, double* P, int ldp, const double* WK, Ops&& ops) {
    for (int pass = ws.planned.count(key) ? 1 : 0; pass < 2; ++pass) {
        Rec rec{ws, F, ld, X, ldx, P, ldp, pass == 0};
        ops(rec, WK);
        if (ws.err != hipSuccess) return;
    }
    ws.planned.insert(key);
}

}  // namespace

void sym_inverse_async(InvWorkspace& ws, double* F, uint32_t n, uint32_t np, bool scale_to_unity, bool reset_info) {
    if (reset_info) info_reset(ws);  // 0x7f7f7f7f = "no failure" sentinel for atomicMin
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
    run_products(ws, plan_key(DK_INVERSE, 0, T, 0), F, (int)np, ws.X, (int)np, ws.W, (int)np, nullptr, ops);     // (tables first: the blocking uploads never sit between kernels)
    gemm_profile_close(ws);
    if (scale_to_unity) launch_scale_sym(F, ws.svec, n, np, 0, ws.stream);
    inv_note_error(ws, hipGetLastError(), "inverse: scaling launch");
    inv_note_error(ws, hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, ws.stream), "info copy");
}

double schur_split() {
    return 0.2;
}

void sym_schur_keep_async(InvWorkspace& ws, double* F, double* X, int ld, int ti, int tj) {
    info_reset(ws);
    auto ops = [&](Rec& rec, const double*) {
        if (ti > 0) rec.node(0, ti);
        if (ti > 0 && tj > 0) rec.eliminate(0, ti, tj);
    };
    run_products(ws, plan_key(DK_SCHUR_KEEP, 0, ti, tj), F, ld, X, ld, ws.W, ld, nullptr, ops);
    gemm_profile_close(ws);
}

void sym_complete_async(InvWorkspace& ws, double* F, double* X, int ld, const double* WK, int ldwk, int ti, int tj, int what) {
    if (what & 1) info_reset(ws);
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
    run_products(ws, plan_key(DK_COMPLETE, what & 3, ti, tj), F, ld, X, ld, ws.W, ld, WK, ops);
    gemm_profile_close(ws);
}

void sym_spine_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj) {
    info_reset(ws);
    const double split = schur_split();
    auto ops = [&](Rec& rec, const double*) { rec.spine(ti, tj, split); };
    run_products(ws, plan_key(DK_SPINE, 0, ti, tj), F, ld, S, ld, ws.W, ld, nullptr, ops);
    gemm_profile_close(ws);
}

void sym_spine_kept_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj) {
    info_reset(ws);
    auto ops = [&](Rec& rec, const double*) {
        Rec::PLocal pl(rec, ti);
        rec.node(ti, tj);
    };
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
    info_reset(ws);
    const int ldx = ti * 128;
    const double split = schur_split();
    auto ops = [&](Rec& rec, const double*) { rec.schur(ti, tj, split); };
    run_products(ws, plan_key(DK_SCHUR, 0, ti, tj), F, ld, ws.X, ldx, P, ldp, nullptr, ops);
    gemm_profile_close(ws);
}

}  // namespace dnagpu
