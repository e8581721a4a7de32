// Implementation of the C-ABI declared in include/dnagpu.h.
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>
#include "adjust_kernels.h"
#include "small_steps.h"
#include "ctx.h"
#include "la_kernels.h"
#include "terrestrial.h"
#include "sym_inverse.h"

using namespace dnagpu;

// dnagpu_schur_carry leaves its junction matrix in information form (no inverse of the complement): default;
// dnagpu_debug_set_info_carry(0) gives the estimates form (complement inverted, estimates + corrections), e.g. to compare the two
static std::atomic<int> g_info_carry{1};

namespace {

constexpr uint32_t SYMV_CHUNKS = 32;
constexpr int INFO_SENTINEL = 0x7f7f7f7f;
// dnagpu_debug_fail_batch_workspaces(n): the next n allocations of a batch's member workspaces fail as if HBM were full (tests)
std::atomic<long> g_fail_batch_ws{0};
static_assert(DNAGPU_CHAIN_BATCH_MAX == BATCH_MAX && DNAGPU_BATCH_MAX <= BATCH_MAX, "include/dnagpu.h and la_kernels.h disagree on the batch sizes");
constexpr int BLOCK_BATCH_MAX = DNAGPU_BATCH_MAX;      // (the by-value member tables of the block batches: adjust_kernels.h FormBatch ...)

// Error text and dpotrf-style info are kept twice: per host thread (every chain of a context is driven by its own host thread,
// and two chains may fail together) and in the context, under a mutex, for any other thread that asks afterwards.
thread_local std::string tls_err;
thread_local int tls_info = 0;
thread_local const dnagpu_ctx* tls_ctx = nullptr;
thread_local bool tls_unread = false;

void note_error(dnagpu_ctx* ctx, const char* text, int info) {
    tls_err = text;
    tls_info = info;
    tls_ctx = ctx;
    tls_unread = true;
    std::lock_guard<std::mutex> lk(ctx->err_mutex);
    ctx->err = text;
    ctx->last_info = info;
}

int fail(dnagpu_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
    if (ctx) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof(buf), "%s", what);
        note_error(ctx, buf, 0);
    }
    return code;
}

#define HIPCHK(call)                                                      \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, #call, e_); \
    } while (0)

#define CHK_CTX()                                   \
    do {                                            \
        if (!ctx) return DNAGPU_EINVAL;             \
        hipError_t e0_ = hipSetDevice(ctx->device); \
        if (e0_ != hipSuccess) return fail(ctx, DNAGPU_EHIP, "hipSetDevice", e0_); \
    } while (0)

#define CHK_CHAIN()                                                               \
    do {                                                                          \
        if (chain < 0 || chain >= DNAGPU_NUM_CHAINS) return fail(ctx, DNAGPU_EINVAL, "bad chain"); \
    } while (0)

int ensure_ws(dnagpu_ctx* ctx, int chain, uint32_t np) {
    InvWorkspace& ws = ctx->ws[chain];
    if (ws.np_cap >= np) return DNAGPU_OK;
    bool prof = ws.prof.enabled;
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    inv_workspace_free(ws);
    hipError_t e = inv_workspace_alloc(ws, np, ctx->stream[chain]);
    if (e != hipSuccess) {
        inv_workspace_free(ws);
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "inverse workspace allocation", e);
    }
    ws.prof.enabled = prof;
    ws.dist_rank = ctx->dist_rank;
    ws.dist_world = ctx->dist_world;
    ws.exchange = ctx->exchange;
    ws.exchange_user = ctx->exchange_user;
    return DNAGPU_OK;
}

int ensure_symv(dnagpu_ctx* ctx, int chain, uint32_t np) {
    if (ctx->symv_cap[chain] >= np) return DNAGPU_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    if (ctx->symv_part[chain]) hipFree(ctx->symv_part[chain]);
    ctx->symv_part[chain] = nullptr;
    ctx->symv_cap[chain] = 0;
    HIPCHK(dnagpu::poison_malloc(&ctx->symv_part[chain], (size_t)SYMV_CHUNKS * np * sizeof(double)));
    ctx->symv_cap[chain] = np;
    return DNAGPU_OK;
}

int ensure_scr_u32(dnagpu_ctx* ctx, int chain, size_t count) {
    if (ctx->scr_u32_cap[chain] >= count) return DNAGPU_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    if (ctx->scr_u32[chain]) hipFree(ctx->scr_u32[chain]);
    ctx->scr_u32[chain] = nullptr;
    ctx->scr_u32_cap[chain] = 0;
    size_t cap = std::max<size_t>(count, 4096);
    HIPCHK(dnagpu::poison_malloc(&ctx->scr_u32[chain], cap * sizeof(uint32_t)));
    ctx->scr_u32_cap[chain] = cap;
    return DNAGPU_OK;
}

int ensure_scr_f64(dnagpu_ctx* ctx, int chain, size_t count) {
    if (ctx->scr_f64_cap[chain] >= count) return DNAGPU_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    if (ctx->scr_f64[chain]) hipFree(ctx->scr_f64[chain]);
    ctx->scr_f64[chain] = nullptr;
    ctx->scr_f64_cap[chain] = 0;
    size_t cap = std::max<size_t>(count, 4096);
    HIPCHK(dnagpu::poison_malloc(&ctx->scr_f64[chain], cap * sizeof(double)));
    ctx->scr_f64_cap[chain] = cap;
    return DNAGPU_OK;
}

// ---- HBM-side roofline: events around the launches of the large bandwidth-bound kernels (dnagpu_profile_hbm_enable) -------------------
// The bracketed duration is the kernel's own only when nothing else runs on the device: bench.py collects these in its one-chain step.
struct HbmTimed {
    dnagpu_ctx* ctx;
    int chain;
    bool on;
    dnagpu_ctx::HbmRec rec;
    HbmTimed(dnagpu_ctx* c, int ch, int kind, double bytes) : ctx(c), chain(ch), on(c->hbm_profile && bytes >= 1.0e6) {
        if (!on) return;
        rec.kind = kind;
        rec.bytes = bytes;
        rec.e0 = rec.e1 = nullptr;
        auto& fr = ctx->hbm_free[chain];
        for (hipEvent_t* e : {&rec.e0, &rec.e1}) {
            if (!fr.empty()) {
                *e = fr.back();
                fr.pop_back();
            } else if (hipEventCreate(e) != hipSuccess) {
                (void)hipGetLastError();
                *e = nullptr;
            }
        }
        if (!rec.e0 || !rec.e1) {
            on = false;
            return;
        }
        gemm_profile_close(ctx->ws[chain]);
        (void)hipEventRecord(rec.e0, ctx->stream[chain]);
    }
    ~HbmTimed() {
        if (!on) return;
        (void)hipEventRecord(rec.e1, ctx->stream[chain]);
        ctx->hbm_recs[chain].push_back(rec);
    }
};

void free_index_cache(dnagpu_ctx* ctx, int chain) {
    for (auto& kv : ctx->idx_cache[chain])
        if (kv.second.dev) hipFree(kv.second.dev);
    ctx->idx_cache[chain].clear();
    for (auto& kv : ctx->val_cache[chain])
        if (kv.second.dev) hipFree(kv.second.dev);
    ctx->val_cache[chain].clear();
}

// upload a small host array to the chain's staging buffer (stream ordered).  Lists of 64 entries or more are kept (ctx.h IndexList):
// the chain steps and rigorous solves of the condensed schedule send the same station lists in every iteration, and the upload's
// stream synchronisation -- 4 - 5 per chain step of ~1.8 ms -- was a tenth of the chain phase
int stage_u32(dnagpu_ctx* ctx, int chain, const uint32_t* host, size_t count, uint32_t** dev) {
    // (round 5: lists from 4 entries on -- a dnasegment-default cut has junction lists of a few dozen stations, and every list that misses
    //  the cache costs a stream synchronisation in every chain step of every iteration)
    if (count >= 1) {      // (every list: a one-station list of constraints through the staging buffer kept its whole batch off the merged launches)
        uint64_t h = 1469598103934665603ull ^ (uint64_t)count;
        for (size_t i = 0; i < count; ++i) h = (h ^ host[i]) * 1099511628211ull;
        auto& cache = ctx->idx_cache[chain];
        auto range = cache.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second.host.size() == count && !memcmp(it->second.host.data(), host, count * sizeof(uint32_t))) {
                *dev = it->second.dev;
                return DNAGPU_OK;
            }
        // a handful of lists per block and chain step: the cap follows the block count (cfg4: 128 blocks + 128 condensed blocks walked by
        // one chain), and a full cache is NOT flushed -- a flush in every iteration would cost more than the staging buffer it replaced --
        // the list at hand simply takes the staging buffer below
        const size_t cap = std::max<size_t>(512, 24 * ctx->blocks.size());
        uint32_t* d = nullptr;
        hipError_t e = cache.size() >= cap ? hipErrorOutOfMemory : dnagpu::poison_malloc(&d, count * sizeof(uint32_t));
        if (e == hipSuccess) {
            e = hipMemcpy(d, host, count * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                dnagpu_ctx::IndexList l;
                l.host.assign(host, host + count);
                l.dev = d;
                cache.emplace(h, std::move(l));
                *dev = d;
                return DNAGPU_OK;
            }
            hipFree(d);
        }
        (void)hipGetLastError();        // (no room for a copy of its own: the staging buffer as before)
    }
    int rc = ensure_scr_u32(ctx, chain, count);
    if (rc) return rc;
    // the previous user of the staging buffer may still be running: the copy is
    // stream ordered behind it, but the host source must stay valid -> sync copy
    HIPCHK(hipMemcpyAsync(ctx->scr_u32[chain], host, count * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    *dev = ctx->scr_u32[chain];
    return DNAGPU_OK;
}

int stage_f64(dnagpu_ctx* ctx, int chain, const double* host, size_t count, double** dev) {
    // (round 5: kept by content like the index lists -- the constraint weights of a chain step are the same in every iteration, and the
    //  upload's stream synchronisation stood in every step of a chain that otherwise waits for nothing)
    if (count >= 9 && count <= 9 * 4096) {
        uint64_t h = 1469598103934665603ull ^ (uint64_t)count;
        for (size_t i = 0; i < count; ++i) {
            uint64_t bits;
            memcpy(&bits, host + i, sizeof(bits));
            h = (h ^ bits) * 1099511628211ull;
        }
        auto& cache = ctx->val_cache[chain];
        auto range = cache.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second.host.size() == count && !memcmp(it->second.host.data(), host, count * sizeof(double))) {
                *dev = it->second.dev;
                return DNAGPU_OK;
            }
        const size_t cap = std::max<size_t>(512, 24 * ctx->blocks.size());
        double* d = nullptr;
        hipError_t e = cache.size() >= cap ? hipErrorOutOfMemory : dnagpu::poison_malloc(&d, count * sizeof(double));
        if (e == hipSuccess) {
            e = hipMemcpy(d, host, count * sizeof(double), hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                dnagpu_ctx::ValueList l;
                l.host.assign(host, host + count);
                l.dev = d;
                cache.emplace(h, std::move(l));
                *dev = d;
                return DNAGPU_OK;
            }
            hipFree(d);
        }
        (void)hipGetLastError();
    }
    int rc = ensure_scr_f64(ctx, chain, count);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->scr_f64[chain], host, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    *dev = ctx->scr_f64[chain];
    return DNAGPU_OK;
}

Block* find_block(dnagpu_ctx* ctx, uint32_t blk) {
    auto it = ctx->blocks.find(blk);
    return it == ctx->blocks.end() ? nullptr : &it->second;
}

void free_block(Block& b) {
    // (dnagpu_block_create's arrays -- stations, vectors per chain, baselines -- are one arena)
    if (b.arena) hipFree(b.arena);
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c)
        for (void* p : {(void*)b.tb[c], (void*)b.trow[c], b.wb_own ? (void*)b.wb[c] : nullptr})
            if (p) hipFree(p);
    void* ptrs[] = {b.Wblk, b.pair_row, b.pair_col, b.pair_off, b.pair_ent, b.inc_off, b.inc,
                    b.t_type, b.t_stn, b.t_blk0, b.t_vec0, b.t_val, b.t_pre, b.t_var, b.t_ih, b.t_th, b.s_llh, b.s_geoid, b.s_defl,
                    b.ds_a, b.ds_b, b.ds_pq, b.ds_w, b.ds_row0, b.ds_k, b.ds_woff, b.ds_wts, b.osc_gidx, b.osc_visit, b.corr_keep, b.schur_idx[0], b.schur_idx[1], b.schur_map[0], b.schur_map[1], b.schur_spos[0], b.schur_spos[1]};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (void* p : b.retired) hipFree(p);
    b = Block();
}

double* station_vec(Block& b, int which, int chain) {
    switch (which) {
        case 0: return b.x_orig;
        case 1: return b.x_est[chain];
        case 2: return b.x_rig;
        default: return nullptr;
    }
}

int check_info_batch(dnagpu_ctx* ctx, int chain, int nb, int* failed_member);

int check_info(dnagpu_ctx* ctx, int chain) {
    // an enqueue that failed (table allocation, launch, copy) leaves `info` at its sentinel: it is reported first
    const char* where = nullptr;
    hipError_t e = inv_take_error(ctx->ws[chain], &where);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, where ? where : "inverse", e);
    if ((e = hipGetLastError()) != hipSuccess) return fail(ctx, DNAGPU_EHIP, "kernel launch", e);   // (launches of this thread)
    int info = *ctx->ws[chain].info_host;
    if (info != INFO_SENTINEL) {
        char buf[128];
        snprintf(buf, sizeof(buf), "Matrix inversion failed, the matrix is singular. (leading minor %d)", info);
        note_error(ctx, buf, info);
        return DNAGPU_ENOTPOSDEF;
    }
    tls_info = 0;
    return DNAGPU_OK;
}

// HIP serialises streams that share one of its GPU_MAX_HW_QUEUES (default 4) hardware queues; the four chain streams fill them,
// and RCCL's or the staged mode's copy streams beside them made two chains share a queue.  Raised
// when the library is loaded -- effective unless the process started the HIP runtime before, or set the variable itself.
// Opt-in (DNAGPU_SET_HW_QUEUES=1): a drop-in library does not change its host's environment by itself; the host sets
// GPU_MAX_HW_QUEUES=16 before the HIP runtime starts (bench.py and the test harness do).
__attribute__((constructor)) void dnagpu_hw_queues() {
    const char* e = getenv("DNAGPU_SET_HW_QUEUES");
    if (e && atoi(e) != 0) setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

}  // namespace

extern "C" {

int dnagpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dnagpu_create(int device, dnagpu_ctx** out) {
    if (!out) return DNAGPU_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DNAGPU_ENODEVICE;
    dnagpu_ctx* ctx = new (std::nothrow) dnagpu_ctx();
    if (!ctx) return DNAGPU_ENOMEM;
    ctx->device = device;
    ctx->info_carry = g_info_carry.load();
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return DNAGPU_EHIP;
    }
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        if (hipStreamCreateWithFlags(&ctx->stream[c], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev[c], hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc(&ctx->red_val_host[c], sizeof(double)) != hipSuccess ||
            hipHostMalloc(&ctx->red_idx_host[c], sizeof(uint32_t)) != hipSuccess) {
            dnagpu_destroy(ctx);
            return DNAGPU_EHIP;
        }
        ctx->ws[c].stream = ctx->stream[c];
    }
    if (dnagpu::poison_malloc(&ctx->bad_dev, sizeof(int)) != hipSuccess) {
        dnagpu_destroy(ctx);
        return DNAGPU_ENOMEM;
    }
    *out = ctx;
    return DNAGPU_OK;
}

void dnagpu_destroy(dnagpu_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (auto& kv : ctx->blocks) free_block(kv.second);
    ctx->blocks.clear();
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        inv_workspace_free(ctx->ws[c]);
        if (ctx->symv_part[c]) hipFree(ctx->symv_part[c]);
        if (ctx->scr_u32[c]) hipFree(ctx->scr_u32[c]);
        free_index_cache(ctx, c);
        if (ctx->scr_f64[c]) hipFree(ctx->scr_f64[c]);
        if (ctx->red_val_host[c]) hipHostFree(ctx->red_val_host[c]);
        if (ctx->red_idx_host[c]) hipHostFree(ctx->red_idx_host[c]);
        if (ctx->ev[c]) hipEventDestroy(ctx->ev[c]);
        if (ctx->stream[c]) hipStreamDestroy(ctx->stream[c]);
        if (ctx->stage_buf[c]) hipFree(ctx->stage_buf[c]);
        if (ctx->pack_done[c]) hipEventDestroy(ctx->pack_done[c]);
        if (ctx->copy_done[c]) hipEventDestroy(ctx->copy_done[c]);
        if (ctx->copy_stream[c]) hipStreamDestroy(ctx->copy_stream[c]);
    }
    if (ctx->bad_dev) hipFree(ctx->bad_dev);
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c)
        if (ctx->plan_scratch[c]) hipFree(ctx->plan_scratch[c]);
    for (void* p : {(void*)ctx->osc_prev, (void*)ctx->osc_seen, (void*)ctx->osc_cnt, (void*)ctx->osc_flagged, ctx->osc_rows, (void*)ctx->osc_off, ctx->osc_visits})
        if (p) hipFree(p);
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        for (auto& r : ctx->hbm_recs[c]) {
            hipEventDestroy(r.e0);
            hipEventDestroy(r.e1);
        }
        for (hipEvent_t e : ctx->hbm_free[c]) hipEventDestroy(e);
    }
    delete ctx;
}

// the calling thread's own failure on this context if it has one it has not read yet, else the context's latest
const char* dnagpu_last_error(const dnagpu_ctx* ctx) {
    if (!ctx) return "null context";
    if (!(tls_ctx == ctx && tls_unread)) {
        dnagpu_ctx* c = const_cast<dnagpu_ctx*>(ctx);
        std::lock_guard<std::mutex> lk(c->err_mutex);
        tls_err = c->err;
        tls_info = c->last_info;
        tls_ctx = ctx;
    }
    tls_unread = false;
    return tls_err.c_str();
}
int dnagpu_last_info(const dnagpu_ctx* ctx) {
    if (!ctx) return 0;
    if (tls_ctx == ctx) return tls_info;
    dnagpu_ctx* c = const_cast<dnagpu_ctx*>(ctx);
    std::lock_guard<std::mutex> lk(c->err_mutex);
    return c->last_info;
}

int dnagpu_host_alloc(dnagpu_ctx* ctx, size_t bytes, void** out) {
    CHK_CTX();
    if (!out) return fail(ctx, DNAGPU_EINVAL, "host_alloc: null out");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 8);
    if (e != hipSuccess) return fail(ctx, DNAGPU_ENOMEM, "page-locked host allocation", e);
    return DNAGPU_OK;
}
void dnagpu_host_free(dnagpu_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) hipSetDevice(ctx->device);
    hipHostFree(p);
}

int dnagpu_device_alloc(dnagpu_ctx* ctx, size_t bytes, void** out) {
    CHK_CTX();
    if (!out) return fail(ctx, DNAGPU_EINVAL, "device_alloc: null out");
    *out = nullptr;
    hipError_t e = dnagpu::poison_malloc(out, bytes ? bytes : 8);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "device allocation", e);
    }
    return DNAGPU_OK;
}
void dnagpu_device_free(dnagpu_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) hipSetDevice(ctx->device);
    hipFree(p);
}
int dnagpu_copy(dnagpu_ctx* ctx, void* dst, const void* src, size_t bytes) {
    CHK_CTX();
    if (bytes && (!dst || !src)) return fail(ctx, DNAGPU_EINVAL, "copy: null pointer");
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return DNAGPU_OK;
}

int dnagpu_mem_info(dnagpu_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    CHK_CTX();
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return DNAGPU_OK;
}

int dnagpu_sync(dnagpu_ctx* ctx) {
    CHK_CTX();
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) HIPCHK(hipStreamSynchronize(ctx->stream[c]));
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c)
        if (ctx->copy_stream[c]) {
            HIPCHK(hipStreamSynchronize(ctx->copy_stream[c]));
            ctx->copy_pending[c] = false;
        }
    return DNAGPU_OK;
}

int dnagpu_chain_sync(dnagpu_ctx* ctx, int chain) {
    CHK_CTX();
    CHK_CHAIN();
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_chain_wait(dnagpu_ctx* ctx, int waiter, int signaller) {
    CHK_CTX();
    if (waiter < 0 || waiter >= DNAGPU_NUM_CHAINS || signaller < 0 || signaller >= DNAGPU_NUM_CHAINS)
        return fail(ctx, DNAGPU_EINVAL, "bad chain");
    if (waiter == signaller) return DNAGPU_OK;
    HIPCHK(hipEventRecord(ctx->ev[signaller], ctx->stream[signaller]));
    HIPCHK(hipStreamWaitEvent(ctx->stream[waiter], ctx->ev[signaller], 0));
    return DNAGPU_OK;
}

/* ---- profiling ------------------------------------------------------------ */
int dnagpu_debug_fail_allocation(long nth) {
    dnagpu::fault_inject_reset(nth);
    return DNAGPU_OK;
}

int dnagpu_debug_fail_batch_workspaces(long n) {
    g_fail_batch_ws.store(n < 0 ? 0 : n);
    return DNAGPU_OK;
}

long dnagpu_debug_set_small_tiles(long tiles) { return dnagpu::small_tiles_set(tiles); }
long dnagpu_debug_set_tiny_tiles(long tiles) { return dnagpu::tiny_tiles_set(tiles); }
int dnagpu_debug_set_info_carry(int on) { return g_info_carry.exchange(on ? 1 : 0); }
int dnagpu_ctx_set_info_carry(dnagpu_ctx* ctx, int on) {
    if (!ctx) return DNAGPU_EINVAL;
    const int old = ctx->info_carry;
    ctx->info_carry = on ? 1 : 0;
    return old;
}
int dnagpu_info_carry(const dnagpu_ctx* ctx) { return ctx ? ctx->info_carry : g_info_carry.load(); }

long dnagpu_debug_tile_order(int mt, int nt, int K, int kmode, int lower, int tile, int jt_lo, int jt_hi, uint32_t* out, long cap, int* per_workgroup) {
    std::vector<uint32_t> t = dnagpu::build_tile_order(mt, nt, K, kmode, lower, tile, jt_lo, jt_hi);
    if (per_workgroup) *per_workgroup = 1;
    for (long i = 0; out && i < cap && i < (long)t.size(); ++i) out[i] = t[i];
    return (long)t.size();
}

int dnagpu_profile_enable(dnagpu_ctx* ctx, int on) {
    CHK_CTX();
    ctx->profile = on != 0;
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) ctx->ws[c].prof.enabled = ctx->profile;
    return DNAGPU_OK;
}
int dnagpu_profile_reset(dnagpu_ctx* ctx) {
    CHK_CTX();
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        HIPCHK(hipStreamSynchronize(ctx->stream[c]));
        gemm_profile_reset(ctx->ws[c]);
    }
    ctx->profile_ms_acc = 0.0;
    return DNAGPU_OK;
}
int dnagpu_profile_hbm_enable(dnagpu_ctx* ctx, int on) {
    CHK_CTX();
    ctx->hbm_profile = on != 0;
    return DNAGPU_OK;
}

int dnagpu_profile_hbm_get(dnagpu_ctx* ctx, double bytes[8], double ms[8], uint64_t count[8], int reset) {
    CHK_CTX();
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        if (ctx->hbm_recs[c].empty()) continue;
        HIPCHK(hipStreamSynchronize(ctx->stream[c]));
        for (const dnagpu_ctx::HbmRec& r : ctx->hbm_recs[c]) {
            float dt = 0.f;
            if (hipEventElapsedTime(&dt, r.e0, r.e1) == hipSuccess && r.kind >= 0 && r.kind < 8) {
                ctx->hbm_bytes[r.kind] += r.bytes;
                ctx->hbm_ms[r.kind] += dt;
                ctx->hbm_count[r.kind]++;
            }
            ctx->hbm_free[c].push_back(r.e0);
            ctx->hbm_free[c].push_back(r.e1);
        }
        ctx->hbm_recs[c].clear();
    }
    for (int k = 0; k < 8; ++k) {
        if (bytes) bytes[k] = ctx->hbm_bytes[k];
        if (ms) ms[k] = ctx->hbm_ms[k];
        if (count) count[k] = ctx->hbm_count[k];
        if (reset) {
            ctx->hbm_bytes[k] = ctx->hbm_ms[k] = 0.0;
            ctx->hbm_count[k] = 0;
        }
    }
    return DNAGPU_OK;
}

int dnagpu_profile_get(dnagpu_ctx* ctx, double* gemm_flops, double* gemm_ms, uint64_t* launches) {
    CHK_CTX();
    // gemm_ms = length of the UNION of the timed GEMM runs of all chains: with one chain this is the plain sum of
    // the run durations; with two chains (multi-thread mode) runs of the two streams overlap in time and share the
    // machine, and summing them would count that time twice.
    double f = 0;
    uint64_t l = 0;
    hipEvent_t base = nullptr;
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        gemm_profile_close(ctx->ws[c]);
        HIPCHK(hipStreamSynchronize(ctx->stream[c]));
        if (!base && ctx->ws[c].prof.used) base = ctx->ws[c].prof.pool[0];
    }
    std::vector<std::pair<float, float>> iv;
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        GemmProfile& p = ctx->ws[c].prof;
        for (size_t i = 0; i + 1 < p.used; i += 2) {
            float t0 = 0.f, dt = 0.f;
            hipEventElapsedTime(&t0, base, p.pool[i]);
            hipEventElapsedTime(&dt, p.pool[i], p.pool[i + 1]);
            iv.emplace_back(t0, t0 + dt);
        }
        p.used = 0;
        f += p.flops;
        l += p.launches;
    }
    std::sort(iv.begin(), iv.end());
    double ms = 0.0;
    float cur_end = -1e30f;
    for (const auto& x : iv) {
        if (x.first >= cur_end) {
            ms += x.second - x.first;
            cur_end = x.second;
        } else if (x.second > cur_end) {
            ms += x.second - cur_end;
            cur_end = x.second;
        }
    }
    ctx->profile_ms_acc += ms;
    if (gemm_flops) *gemm_flops = f;
    if (gemm_ms) *gemm_ms = ctx->profile_ms_acc;
    if (launches) *launches = l;
    return DNAGPU_OK;
}

/* ---- matrices -------------------------------------------------------------- */
int dnagpu_matrix_create(dnagpu_ctx* ctx, uint32_t n_max, dnagpu_matrix** out) {
    CHK_CTX();
    if (!out) return fail(ctx, DNAGPU_EINVAL, "null out");
    *out = nullptr;
    dnagpu_matrix* m = new (std::nothrow) dnagpu_matrix();
    if (!m) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    m->n_max = n_max;
    m->np_max = pad128(n_max);
    // one spare tile row: dnagpu_schur_carry keeps (np + 128) x np panels here
    hipError_t e = dnagpu::poison_malloc(&m->F, ((size_t)m->np_max + 128) * m->np_max * sizeof(double));
    if (e == hipSuccess) e = dnagpu::poison_malloc(&m->jest, (size_t)m->np_max * sizeof(double));
    if (e != hipSuccess) {
        if (m->F) hipFree(m->F);
        delete m;
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "matrix allocation", e);
    }
    m->n = 0;
    m->np = 128;
    *out = m;
    return DNAGPU_OK;
}

void dnagpu_matrix_destroy(dnagpu_ctx* ctx, dnagpu_matrix* m) {
    if (!m) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
    }
    if (m->F) hipFree(m->F);
    if (m->jest) hipFree(m->jest);
    if (m->jrhs) hipFree(m->jrhs);
    delete m;
}

int dnagpu_matrix_reset(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, uint32_t n) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || n > m->n_max) return fail(ctx, DNAGPU_EINVAL, "matrix_reset: order exceeds capacity");
    m->n = n;
    m->np = pad128(n);
    launch_init_padded(m->F, n, m->np, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_matrix_upload_packed(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* ap, uint32_t n) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || n > m->n_max || (!ap && n)) return fail(ctx, DNAGPU_EINVAL, "matrix_upload_packed: bad arguments");
    m->n = n;
    m->np = pad128(n);
    size_t cnt = (size_t)n * (n + 1) / 2;
    // stage the packed data in the inverse workspace's X buffer (np^2 >= n(n+1)/2)
    int rc = ensure_ws(ctx, chain, m->np);
    if (rc) return rc;
    if (cnt) HIPCHK(hipMemcpyAsync(ctx->ws[chain].X, ap, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream[chain]));
    launch_unpack_lower(ctx->ws[chain].X, m->F, n, m->np, ctx->stream[chain]);
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_matrix_download_packed(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || (!ap && m->n)) return fail(ctx, DNAGPU_EINVAL, "matrix_download_packed: bad arguments");
    size_t cnt = (size_t)m->n * (m->n + 1) / 2;
    int rc = ensure_ws(ctx, chain, m->np);
    if (rc) return rc;
    launch_pack_lower(m->F, ctx->ws[chain].X, m->n, m->np, ctx->stream[chain]);
    if (cnt) HIPCHK(hipMemcpyAsync(ap, ctx->ws[chain].X, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

// the chain's copy stream, its two events and a staging buffer of at least cnt doubles; 1: made, 0: not (the caller takes the blocking path)
static int ensure_copy_stage(dnagpu_ctx* ctx, int chain, size_t cnt, int* rc_out) {
    *rc_out = DNAGPU_OK;
    if (!ctx->copy_stream[chain]) {
        hipStream_t cs = nullptr;
        hipEvent_t pd = nullptr, cd = nullptr;
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&pd, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&cd, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (cd) hipEventDestroy(cd);
            if (pd) hipEventDestroy(pd);
            if (cs) hipStreamDestroy(cs);
            return 0;
        }
        ctx->copy_stream[chain] = cs;
        ctx->pack_done[chain] = pd;
        ctx->copy_done[chain] = cd;
    }
    if (ctx->stage_cap[chain] < cnt) {
        hipError_t e = hipStreamSynchronize(ctx->copy_stream[chain]);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream[chain]);      // (an upload through the buffer that the chain has not consumed yet)
        if (e != hipSuccess) {
            *rc_out = fail(ctx, DNAGPU_EHIP, "copy staging buffer", e);
            return 0;
        }
        ctx->copy_pending[chain] = false;
        if (ctx->stage_buf[chain]) hipFree(ctx->stage_buf[chain]);
        ctx->stage_buf[chain] = nullptr;
        ctx->stage_cap[chain] = 0;
        if (dnagpu::poison_malloc(&ctx->stage_buf[chain], cnt * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->stage_buf[chain] = nullptr;
            return 0;
        }
        ctx->stage_cap[chain] = cnt;
    }
    return 1;
}

// A light factor to page-locked HOST memory as its packed lower triangle, and back (round 6): a block the HBM budget denies a kept factor,
// and whose packed variance matrix will live in a host slot of the staged store, parks its factor THERE between its condensing step and its
// variance matrix -- 2.9 GB at n = 27 000 take 51 ms over the host link, beside the other chains' work, where eliminating the block again
// takes 100 ms of the whole chip.  Out: packed on the chain's stream into the chain's staging buffer, copied on its copy stream
// (dnagpu_copies_sync before the slot is read).  Back: copied into the staging buffer and unpacked on the chain's stream.
int dnagpu_partial_pack_host_async(dnagpu_ctx* ctx, int chain, const dnagpu_partial* p, double* host_ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!p || !host_ap || !p->spine || !p->valid || !p->npp) return fail(ctx, DNAGPU_EINVAL, "partial_pack_host_async: no light factor to pack");
    const size_t cnt = (size_t)p->npp * (p->npp + 1) / 2;
    int rc = DNAGPU_OK;
    if (!ensure_copy_stage(ctx, chain, cnt, &rc)) {
        if (rc) return rc;
        // no copy stream or no room for a staging buffer: through the chain's L^-1 workspace, the chain waits for its copy
        rc = ensure_ws(ctx, chain, p->npp);
        if (rc) return rc;
        launch_pack_lower(p->X, ctx->ws[chain].W, p->npp, p->npp, ctx->stream[chain]);
        HIPCHK(hipMemcpyAsync(host_ap, ctx->ws[chain].W, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream[chain]));
        HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
        return DNAGPU_OK;
    }
    if (ctx->copy_pending[chain]) HIPCHK(hipStreamWaitEvent(ctx->stream[chain], ctx->copy_done[chain], 0));
    launch_pack_lower(p->X, ctx->stage_buf[chain], p->npp, p->npp, ctx->stream[chain]);
    HIPCHK(hipEventRecord(ctx->pack_done[chain], ctx->stream[chain]));
    HIPCHK(hipStreamWaitEvent(ctx->copy_stream[chain], ctx->pack_done[chain], 0));
    HIPCHK(hipMemcpyAsync(host_ap, ctx->stage_buf[chain], cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream[chain]));
    HIPCHK(hipEventRecord(ctx->copy_done[chain], ctx->copy_stream[chain]));
    ctx->copy_pending[chain] = true;
    return DNAGPU_OK;
}

int dnagpu_partial_unpack_host(dnagpu_ctx* ctx, int chain, dnagpu_partial* dst, const dnagpu_partial* src, const double* host_ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!dst || !src || !host_ap || !dst->spine || !src->spine || !src->npp) return fail(ctx, DNAGPU_EINVAL, "partial_unpack_host: bad arguments");
    const size_t cnt = (size_t)src->npp * (src->npp + 1) / 2;
    int rc = DNAGPU_OK;
    double* stage = nullptr;
    if (ensure_copy_stage(ctx, chain, cnt, &rc)) {
        // (a copy OUT of the staging buffer that is still on its way leaves first)
        if (ctx->copy_pending[chain]) HIPCHK(hipStreamWaitEvent(ctx->stream[chain], ctx->copy_done[chain], 0));
        stage = ctx->stage_buf[chain];
    } else {
        if (rc) return rc;
        rc = ensure_ws(ctx, chain, src->npp);          // (the panels' workspace: dead between driver calls, and not where dst lives)
        if (rc) return rc;
        stage = ctx->ws[chain].W;
    }
    HIPCHK(hipMemcpyAsync(stage, host_ap, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream[chain]));
    return dnagpu_partial_unpack_device(ctx, chain, dst, src, stage);
}

int dnagpu_matrix_download_packed_async(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || (!ap && m->n)) return fail(ctx, DNAGPU_EINVAL, "matrix_download_packed_async: bad arguments");
    const size_t cnt = (size_t)m->n * (m->n + 1) / 2;
    if (!cnt) return DNAGPU_OK;
    int rc = DNAGPU_OK;
    if (!ensure_copy_stage(ctx, chain, cnt, &rc))       // (no copy stream, or no room for the staging buffer: the chain waits for its copy)
        return rc ? rc : dnagpu_matrix_download_packed(ctx, chain, m, ap);
    // the previous copy out of this buffer must have left before it is packed again
    if (ctx->copy_pending[chain]) HIPCHK(hipStreamWaitEvent(ctx->stream[chain], ctx->copy_done[chain], 0));
    {
        HbmTimed timed(ctx, chain, DNAGPU_HBM_PACK, 8.0 * (double)m->n * m->n);
        launch_pack_lower(m->F, ctx->stage_buf[chain], m->n, m->np, ctx->stream[chain]);
    }
    HIPCHK(hipEventRecord(ctx->pack_done[chain], ctx->stream[chain]));
    HIPCHK(hipStreamWaitEvent(ctx->copy_stream[chain], ctx->pack_done[chain], 0));
    HIPCHK(hipMemcpyAsync(ap, ctx->stage_buf[chain], cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream[chain]));
    HIPCHK(hipEventRecord(ctx->copy_done[chain], ctx->copy_stream[chain]));
    ctx->copy_pending[chain] = true;
    return DNAGPU_OK;
}

int dnagpu_matrix_pack_device(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* dev_ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || (!dev_ap && m->n)) return fail(ctx, DNAGPU_EINVAL, "matrix_pack_device: bad arguments");
    HbmTimed timed(ctx, chain, DNAGPU_HBM_PACK, 8.0 * (double)m->n * m->n);
    if (m->n) launch_pack_lower(m->F, dev_ap, m->n, m->np, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_matrix_unpack_device(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* dev_ap, uint32_t n) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || n > m->n_max || (!dev_ap && n)) return fail(ctx, DNAGPU_EINVAL, "matrix_unpack_device: bad arguments");
    m->n = n;
    m->np = pad128(n);
    launch_unpack_lower(dev_ap, m->F, n, m->np, ctx->stream[chain]);
    return DNAGPU_OK;
}

// A light (spine-form) factor as its packed lower triangle and back: where HBM cannot hold every block's factor as a square, a block whose
// packed variance matrix will live in HBM anyway lends that slot to its factor during the iterations (half the bytes of the square).
int dnagpu_partial_pack_device(dnagpu_ctx* ctx, int chain, const dnagpu_partial* p, double* dev_ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!p || !dev_ap || !p->spine || !p->valid || !p->npp) return fail(ctx, DNAGPU_EINVAL, "partial_pack_device: no light factor to pack");
    launch_pack_lower(p->X, dev_ap, p->npp, p->npp, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_partial_unpack_device(dnagpu_ctx* ctx, int chain, dnagpu_partial* dst, const dnagpu_partial* src, const double* dev_ap) {
    CHK_CTX();
    CHK_CHAIN();
    if (!dst || !src || !dev_ap || !dst->spine || !src->spine || !src->npp)
        return fail(ctx, DNAGPU_EINVAL, "partial_unpack_device: bad arguments");
    if (src->npp > dst->n_cap || src->njp > dst->k_cap) return fail(ctx, DNAGPU_EINVAL, "partial_unpack_device: retained factor capacity");
    hipStream_t st = ctx->stream[chain];
    if (dst != src) {
        dst->n = src->n; dst->nj = src->nj; dst->nip = src->nip; dst->njp = src->njp; dst->npp = src->npp;
        HIPCHK(hipMemcpyAsync(dst->map, src->map, (size_t)src->npp * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
    if (dst->store) dst->store->n = 0;       // (its storage now holds a factor, not a matrix)
    launch_unpack_lower(dev_ap, dst->X, dst->npp, dst->npp, st);
    dst->completed = false;
    dst->factored = false;
    dst->valid = true;
    return DNAGPU_OK;
}

int dnagpu_copies_sync(dnagpu_ctx* ctx) {
    CHK_CTX();
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        if (!ctx->copy_stream[c]) continue;
        HIPCHK(hipStreamSynchronize(ctx->stream[c]));        // (a pack that was only just enqueued)
        HIPCHK(hipStreamSynchronize(ctx->copy_stream[c]));
        ctx->copy_pending[c] = false;
    }
    return DNAGPU_OK;
}

int dnagpu_matrix_copy(dnagpu_ctx* ctx, int chain, dnagpu_matrix* dst, const dnagpu_matrix* src) {
    CHK_CTX();
    CHK_CHAIN();
    if (!dst || !src || src->n > dst->n_max) return fail(ctx, DNAGPU_EINVAL, "matrix_copy: bad arguments");
    dst->n = src->n;
    dst->np = src->np;
    HIPCHK(hipMemcpyAsync(dst->F, src->F, (size_t)src->np * src->np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream[chain]));
    HIPCHK(hipMemcpyAsync(dst->jest, src->jest, (size_t)src->np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream[chain]));
    dst->form = src->form;
    if (src->form == 1) {
        if (!dst->jrhs && dnagpu::poison_malloc(&dst->jrhs, (size_t)dst->np_max * sizeof(double)) != hipSuccess)
            return fail(ctx, DNAGPU_ENOMEM, "matrix_copy: right-hand side of the information form");
        HIPCHK(hipMemcpyAsync(dst->jrhs, src->jrhs, (size_t)src->np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream[chain]));
    }
    return DNAGPU_OK;
}

int dnagpu_matrix_resize(dnagpu_ctx* ctx, dnagpu_matrix* m, uint32_t n) {
    if (!ctx) return DNAGPU_EINVAL;
    if (!m || n > m->n_max) return fail(ctx, DNAGPU_EINVAL, "matrix_resize: bad arguments");
    m->n = n;
    m->np = pad128(n);
    return DNAGPU_OK;
}

int dnagpu_matrix_device_pointers(const dnagpu_matrix* m, double** matrix, double** vector, uint32_t* np) {
    if (!m || m->form == 1) return DNAGPU_EINVAL;     // (the information form has a third part: see dnagpu_matrix_export)
    if (matrix) *matrix = m->F;
    if (vector) *vector = m->jest;
    if (np) *np = m->np;
    return DNAGPU_OK;
}

int dnagpu_matrix_export(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* dst, size_t cap_doubles) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || !dst) return fail(ctx, DNAGPU_EINVAL, "matrix_export: null argument");
    if (m->form == 1)
        return fail(ctx, DNAGPU_EINVAL, "matrix_export: a junction matrix in information form (dnagpu_schur_carry) has no payload form; "
                                        "use dnagpu_junction_export");
    size_t need = (size_t)m->np * m->np + m->np;
    if (cap_doubles < need) return fail(ctx, DNAGPU_EINVAL, "matrix_export: destination too small");
    HIPCHK(hipMemcpyAsync(dst, m->F, (size_t)m->np * m->np * sizeof(double), hipMemcpyDefault, ctx->stream[chain]));
    HIPCHK(hipMemcpyAsync(dst + (size_t)m->np * m->np, m->jest, (size_t)m->np * sizeof(double), hipMemcpyDefault, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_matrix_import(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* src, uint32_t n) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || !src || n > m->n_max) return fail(ctx, DNAGPU_EINVAL, "matrix_import: bad arguments");
    m->n = n;
    m->np = pad128(n);
    m->form = 0;
    HIPCHK(hipMemcpyAsync(m->F, src, (size_t)m->np * m->np * sizeof(double), hipMemcpyDefault, ctx->stream[chain]));
    HIPCHK(hipMemcpyAsync(m->jest, src + (size_t)m->np * m->np, (size_t)m->np * sizeof(double), hipMemcpyDefault, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

// a junction matrix in either form: matrix, attached estimates, reduced right-hand side (zeros in the estimates form), form
int dnagpu_junction_export(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* dst, size_t cap_doubles) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || !dst) return fail(ctx, DNAGPU_EINVAL, "junction_export: null argument");
    const size_t np = m->np, need = np * np + 2 * np + 1;
    if (cap_doubles < need) return fail(ctx, DNAGPU_EINVAL, "junction_export: destination too small");
    if (m->form == 1 && !m->jrhs) return fail(ctx, DNAGPU_EINVAL, "junction_export: information form without its right-hand side");
    hipStream_t st = ctx->stream[chain];
    const double form = (double)m->form;
    HIPCHK(hipMemcpyAsync(dst, m->F, np * np * sizeof(double), hipMemcpyDefault, st));
    HIPCHK(hipMemcpyAsync(dst + np * np, m->jest, np * sizeof(double), hipMemcpyDefault, st));
    if (m->form == 1)
        HIPCHK(hipMemcpyAsync(dst + np * np + np, m->jrhs, np * sizeof(double), hipMemcpyDefault, st));
    else {
        const std::vector<double> zeros(np, 0.0);       // (dst may be host memory: no memset)
        HIPCHK(hipMemcpyAsync(dst + np * np + np, zeros.data(), np * sizeof(double), hipMemcpyDefault, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipMemcpyAsync(dst + np * np + 2 * np, &form, sizeof(double), hipMemcpyDefault, st));
    HIPCHK(hipStreamSynchronize(st));
    return DNAGPU_OK;
}

int dnagpu_junction_import(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* src, uint32_t n) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || !src || n > m->n_max) return fail(ctx, DNAGPU_EINVAL, "junction_import: bad arguments");
    m->n = n;
    m->np = pad128(n);
    const size_t np = m->np;
    hipStream_t st = ctx->stream[chain];
    double form = 0.0;
    HIPCHK(hipMemcpyAsync(&form, src + np * np + 2 * np, sizeof(double), hipMemcpyDefault, st));
    HIPCHK(hipStreamSynchronize(st));
    if (form != 0.0 && form != 1.0) return fail(ctx, DNAGPU_EINVAL, "junction_import: not a junction payload");
    if (form == 1.0 && !m->jrhs && dnagpu::poison_malloc(&m->jrhs, (size_t)m->np_max * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        m->jrhs = nullptr;
        return fail(ctx, DNAGPU_ENOMEM, "junction_import: right-hand side of the information form");
    }
    HIPCHK(hipMemcpyAsync(m->F, src, np * np * sizeof(double), hipMemcpyDefault, st));
    HIPCHK(hipMemcpyAsync(m->jest, src + np * np, np * sizeof(double), hipMemcpyDefault, st));
    if (form == 1.0) HIPCHK(hipMemcpyAsync(m->jrhs, src + np * np + np, np * sizeof(double), hipMemcpyDefault, st));
    HIPCHK(hipStreamSynchronize(st));
    m->form = (int)form;
    return DNAGPU_OK;
}

int dnagpu_junction_device_pointers(dnagpu_ctx* ctx, dnagpu_matrix* m, int as_form, double** matrix, double** estimates, double** rhs, uint32_t* np,
                                    int* form) {
    CHK_CTX();
    if (!m || as_form > 1) return fail(ctx, DNAGPU_EINVAL, "junction_device_pointers: bad arguments");
    const int f = as_form >= 0 ? as_form : m->form;
    if (f == 1 && !m->jrhs && dnagpu::poison_malloc(&m->jrhs, (size_t)m->np_max * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        m->jrhs = nullptr;
        return fail(ctx, DNAGPU_ENOMEM, "junction_device_pointers: right-hand side of the information form");
    }
    if (as_form >= 0) m->form = as_form;       // (about to receive a junction of that form)
    if (matrix) *matrix = m->F;
    if (estimates) *estimates = m->jest;
    if (rhs) *rhs = f == 1 ? m->jrhs : nullptr;
    if (np) *np = m->np;
    if (form) *form = f;
    return DNAGPU_OK;
}

int dnagpu_set_inverse_exchange(dnagpu_ctx* ctx, int rank, int world, dnagpu_exchange_fn fn, void* user) {
    if (!ctx) return DNAGPU_EINVAL;
    if (world > 1 && fn && (rank < 0 || rank >= world)) return fail(ctx, DNAGPU_EINVAL, "set_inverse_exchange: bad rank");
    ctx->dist_rank = (world > 1 && fn) ? rank : 0;
    ctx->dist_world = (world > 1 && fn) ? world : 1;
    ctx->exchange = (world > 1) ? fn : nullptr;
    ctx->exchange_user = user;
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        InvWorkspace& ws = ctx->ws[c];
        ws.dist_rank = ctx->dist_rank;
        ws.dist_world = ctx->dist_world;
        ws.exchange = ctx->exchange;
        ws.exchange_user = ctx->exchange_user;
    }
    return DNAGPU_OK;
}

int dnagpu_inverse_exchange_stats(dnagpu_ctx* ctx, uint64_t* split_launches, double* bytes_received) {
    if (!ctx) return DNAGPU_EINVAL;
    if (split_launches) *split_launches = ctx->ws[0].split_launches;
    if (bytes_received) *bytes_received = ctx->ws[0].exchanged_bytes;
    return DNAGPU_OK;
}

int dnagpu_invert(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, int scale_to_unity) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m) return fail(ctx, DNAGPU_EINVAL, "invert: null matrix");
    if (m->n == 0) return DNAGPU_OK;
    int rc = ensure_ws(ctx, chain, m->np);
    if (rc) return rc;
    sym_inverse_async(ctx->ws[chain], m->F, m->n, m->np, scale_to_unity != 0);
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return check_info(ctx, chain);
}

/* ---- L2 seam ----------------------------------------------------------------- */
int dnagpu_cholesky_inverse_packed(dnagpu_ctx* ctx, double* ap, uint32_t n, int scale_to_unity) {
    CHK_CTX();
    if (n == 0) return DNAGPU_OK;
    if (!ap) return fail(ctx, DNAGPU_EINVAL, "cholesky_inverse_packed: null matrix");
    if (n == 1) {
        // FormInverseVarianceMatrix special case, dnaadjust.cpp:8474
        ap[0] = 1.0 / ap[0];
        return DNAGPU_OK;
    }
    dnagpu_matrix* m = nullptr;
    int rc = dnagpu_matrix_create(ctx, n, &m);
    if (rc) return rc;
    rc = dnagpu_matrix_upload_packed(ctx, 0, m, ap, n);
    if (!rc) rc = dnagpu_invert(ctx, 0, m, scale_to_unity);
    if (!rc) rc = dnagpu_matrix_download_packed(ctx, 0, m, ap);
    dnagpu_matrix_destroy(ctx, m);
    return rc;
}

int dnagpu_multiply_sym_packed(dnagpu_ctx* ctx, const double* ap, const double* x, double* y, uint32_t n) {
    CHK_CTX();
    if (n == 0) return DNAGPU_OK;
    if (!ap || !x || !y) return fail(ctx, DNAGPU_EINVAL, "multiply_sym_packed: null argument");
    dnagpu_matrix* m = nullptr;
    int rc = dnagpu_matrix_create(ctx, n, &m);
    if (rc) return rc;
    rc = dnagpu_matrix_upload_packed(ctx, 0, m, ap, n);
    double* dx = nullptr;
    if (!rc) rc = ensure_symv(ctx, 0, m->np);
    if (!rc) rc = ensure_scr_f64(ctx, 0, 2 * (size_t)m->np);
    if (!rc) {
        hipStream_t s = ctx->stream[0];
        dx = ctx->scr_f64[0];
        double* dy = dx + m->np;
        // mirror the lower triangle so that the symv kernel can stream full columns
        hipError_t e = hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            launch_symmetrize(m->F, m->n, m->np, s);
            launch_symv(m->F, dx, dy, ctx->symv_part[0], n, m->np, SYMV_CHUNKS, s);
            e = hipMemcpyAsync(y, dy, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(ctx, DNAGPU_EHIP, "multiply_sym_packed", e);
    }
    dnagpu_matrix_destroy(ctx, m);
    return rc;
}

/* ---- blocks ------------------------------------------------------------------ */
int dnagpu_block_create(dnagpu_ctx* ctx, uint32_t blk, uint32_t n_stations, uint32_t n_baselines) {
    CHK_CTX();
    if (find_block(ctx, blk)) {
        int rc = dnagpu_block_destroy(ctx, blk);
        if (rc) return rc;
    }
    Block b;
    b.n_stn = n_stations;
    b.n_bl = n_baselines;
    size_t nv = std::max<size_t>(3 * (size_t)n_stations, 1) * sizeof(double);
    size_t nb = std::max<size_t>(n_baselines, 1);
    // One allocation per block for everything of a fixed size (a dnasegment-default cut of a million stations is 20 000 device blocks:
    // condensed blocks and run systems included -- 56 allocations and 16 memsets each were most of PrepareAdjustment's nine seconds):
    // sizes first, then the pointers into the arena.
    struct Slot { void** p; size_t bytes; };
    std::vector<Slot> slots;
    auto A = [&](void** p, size_t bytes) { slots.push_back({p, bytes}); };
    A((void**)&b.x_orig, nv);
    A((void**)&b.x_rig, nv);
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        A((void**)&b.x_est[c], nv);
        A((void**)&b.rhs[c], nv);
        A((void**)&b.corr[c], nv);
        A((void**)&b.b[c], nb * 3 * sizeof(double));
        A((void**)&b.wb[c], nb * 3 * sizeof(double));
        A((void**)&b.red[c], 2 * sizeof(double));
    }
    A((void**)&b.s1, nb * sizeof(uint32_t));
    A((void**)&b.s2, nb * sizeof(uint32_t));
    A((void**)&b.obs, nb * 3 * sizeof(double));
    A((void**)&b.vec_wrow, nb * sizeof(uint32_t));
    A((void**)&b.vec_c0, nb * sizeof(uint32_t));
    A((void**)&b.vec_k, nb * sizeof(uint32_t));
    size_t total = 0;
    for (const Slot& sl : slots) total += (sl.bytes + 255) & ~(size_t)255;
    hipError_t e = dnagpu::poison_malloc(&b.arena, total);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "block allocation", e);
    }
    {
        uint8_t* at = (uint8_t*)b.arena;
        for (const Slot& sl : slots) {
            *sl.p = at;
            at += (sl.bytes + 255) & ~(size_t)255;
        }
    }
    // (right-hand sides and corrections start at zero; with DNAGPU_POISON_ALLOC the rest keeps its NaN)
    // Each chain's pair -- rhs[c] and corr[c] are neighbours in the arena -- is zeroed by ONE memset on that chain's OWN stream: the chain
    // streams do not wait for each other (hipStreamNonBlocking), and whatever a chain enqueues on the block next is ordered behind it.
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        const size_t span = (size_t)((uint8_t*)b.corr[c] - (uint8_t*)b.rhs[c]) + nv;
        hipError_t em = hipMemsetAsync(b.rhs[c], 0, span, ctx->stream[c]);
        if (em != hipSuccess) {
            (void)hipGetLastError();
            hipDeviceSynchronize();
            hipFree(b.arena);
            return fail(ctx, DNAGPU_EHIP, "block allocation: zeroing", em);
        }
    }
    ctx->osc_key = 0;       // (the visit lists of dnagpu_osc_blocks describe the blocks that existed when they were built)
    ctx->blocks[blk] = b;
    return DNAGPU_OK;
}

int dnagpu_block_destroy(dnagpu_ctx* ctx, uint32_t blk) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "block_destroy: unknown block");
    HIPCHK(hipDeviceSynchronize());
    free_block(*b);
    ctx->blocks.erase(blk);
    ctx->osc_key = 0;       // (dnagpu_osc_blocks: a block of the same id created later may have other stations)
    return DNAGPU_OK;
}

int dnagpu_block_set_stations(dnagpu_ctx* ctx, uint32_t blk, const double* xyz) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b || (!xyz && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_set_stations: bad arguments");
    size_t bytes = 3 * (size_t)b->n_stn * sizeof(double);
    if (!bytes) return DNAGPU_OK;
    // one upload, then one launch that copies it to the originals and to every chain's estimates (ten blocking copies before)
    HIPCHK(hipMemcpy(b->x_rig, xyz, bytes, hipMemcpyHostToDevice));
    launch_reset_block(b->x_rig, b->x_orig, b->x_rig, b->x_est, b->b, DNAGPU_NUM_CHAINS, false, b->s1, b->s2, b->obs, b->n_stn, b->n_bl, ctx->stream[0]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream[0]));
    return DNAGPU_OK;
}

static_assert(DNAGPU_NUM_CHAINS <= 8, "launch_reset_block passes one pointer per chain in its kernel arguments (RESET_MAX_CHAINS)");
int dnagpu_block_reset_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, const double* dev_xyz, int with_b) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || (!dev_xyz && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_reset_stations: bad arguments");
    if (with_b && b->n_t) return fail(ctx, DNAGPU_EINVAL, "block_reset_stations: a block with terrestrial measurements takes dnagpu_block_compute_b");
    launch_reset_block(dev_xyz, b->x_orig, b->x_rig, b->x_est, b->b, DNAGPU_NUM_CHAINS, with_b != 0, b->s1, b->s2, b->obs, b->n_stn, b->n_bl,
                       ctx->stream[chain]);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

void dnagpu_block_table_destroy(dnagpu_ctx* ctx, dnagpu_block_table* t) {
    if (!t) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
    }
    if (t->rows) hipFree(t->rows);
    delete t;
}

int dnagpu_block_table_create(dnagpu_ctx* ctx, uint32_t n, const uint32_t* blks, const int* last, const double* dev_init, const size_t* init_off,
                              dnagpu_block_table** out) {
    CHK_CTX();
    static_assert(DNAGPU_NUM_CHAINS <= 8, "BlockTableRow holds one pointer per chain");
    if (!out || !n || !blks || !last || !dev_init || !init_off) return fail(ctx, DNAGPU_EINVAL, "block_table_create: bad arguments");
    *out = nullptr;
    std::vector<BlockTableRow> host(n);
    uint32_t max_len = 0;
    for (uint32_t q = 0; q < n; ++q) {
        Block* b = find_block(ctx, blks[q]);
        if (!b) return fail(ctx, DNAGPU_EINVAL, "block_table_create: unknown block");
        if (b->n_t || b->n_dsblk) return fail(ctx, DNAGPU_EINVAL, "block_table_create: a block with terrestrial measurements");
        BlockTableRow& r = host[q];
        memset(&r, 0, sizeof(r));
        r.init = dev_init + init_off[q];
        r.x_orig = b->x_orig;
        r.x_rig = b->x_rig;
        for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
            r.x_est[c] = b->x_est[c];
            r.b[c] = b->b[c];
        }
        r.s1 = b->s1;
        r.s2 = b->s2;
        r.obs = b->obs;
        r.n3 = 3 * b->n_stn;
        r.nb3 = 3 * b->n_bl;
        r.last = last[q] ? 1u : 0u;
        max_len = std::max(max_len, std::max(r.n3, r.nb3));
    }
    dnagpu_block_table* t = new (std::nothrow) dnagpu_block_table();
    if (!t) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    if (dnagpu::poison_malloc(&t->rows, (size_t)n * sizeof(BlockTableRow)) != hipSuccess ||
        hipMemcpy(t->rows, host.data(), (size_t)n * sizeof(BlockTableRow), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        dnagpu_block_table_destroy(ctx, t);
        return fail(ctx, DNAGPU_ENOMEM, "block_table_create: rows");
    }
    t->n = n;
    t->max_len = max_len;
    *out = t;
    return DNAGPU_OK;
}

int dnagpu_block_table_apply(dnagpu_ctx* ctx, int chain, const dnagpu_block_table* t, int mode, int chains) {
    CHK_CTX();
    CHK_CHAIN();
    if (!t || (mode != 0 && mode != 1) || chains < 1 || chains > DNAGPU_NUM_CHAINS) return fail(ctx, DNAGPU_EINVAL, "block_table_apply: bad arguments");
    gemm_profile_close(ctx->ws[chain]);
    launch_block_table((const BlockTableRow*)t->rows, t->n, t->max_len, mode, mode == 0 ? DNAGPU_NUM_CHAINS : chains, ctx->stream[chain]);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

int dnagpu_chain_hold_info(dnagpu_ctx* ctx, int chain, int on) {
    CHK_CTX();
    CHK_CHAIN();
    int rc = ensure_ws(ctx, chain, 128);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    if (on && !ws.hold_info) HIPCHK(hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), ctx->stream[chain]));
    ws.hold_info = on != 0;
    return DNAGPU_OK;
}

int dnagpu_chain_take_info(dnagpu_ctx* ctx, int chain) {
    CHK_CTX();
    CHK_CHAIN();
    int rc = ensure_ws(ctx, chain, 128);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    // (every member's word: the run may have held batched steps -- dnagpu_chain_plan_run --, whose member b reports in info[b])
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, BATCH_MAX * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info_batch(ctx, chain, BATCH_MAX, nullptr);
}

int dnagpu_block_set_station_geo(dnagpu_ctx* ctx, uint32_t blk, const double* llh, const double* geoid, const double* defl) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b || (b->n_stn && (!llh || !geoid || !defl))) return fail(ctx, DNAGPU_EINVAL, "block_set_station_geo: bad arguments");
    const size_t ns = std::max<size_t>(b->n_stn, 1);
    if (!b->s_llh) {
        HIPCHK(dnagpu::poison_malloc(&b->s_llh, 3 * ns * sizeof(double)));
        HIPCHK(dnagpu::poison_malloc(&b->s_geoid, ns * sizeof(double)));
        HIPCHK(dnagpu::poison_malloc(&b->s_defl, 2 * ns * sizeof(double)));
    }
    if (!b->n_stn) return DNAGPU_OK;
    HIPCHK(hipMemcpy(b->s_llh, llh, 3 * (size_t)b->n_stn * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->s_geoid, geoid, (size_t)b->n_stn * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->s_defl, defl, 2 * (size_t)b->n_stn * sizeof(double), hipMemcpyHostToDevice));
    return DNAGPU_OK;
}

int dnagpu_block_set_terrestrial(dnagpu_ctx* ctx, uint32_t blk, uint32_t n_t, const char* type, const uint32_t* stn3, const double* value,
                                 const double* pre_adj_meas, const double* variance, const double* inst_height, const double* targ_height,
                                 const uint32_t* cml_pos, const uint32_t* cluster_cml_pos, uint32_t n_clusters) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b || (n_t && (!type || !stn3 || !value || !pre_adj_meas || !variance || !inst_height || !targ_height || !cml_pos)))
        return fail(ctx, DNAGPU_EINVAL, "block_set_terrestrial: bad arguments");
    std::vector<uint32_t> blk0(n_t), vec0(n_t);
    uint32_t nb = 0, nv = 0;
    for (uint32_t t = 0; t < n_t; ++t) {
        if (!dnagpu::tm::is_terrestrial(type[t])) return fail(ctx, DNAGPU_EINVAL, "block_set_terrestrial: measurement type not handled");
        const int ns = dnagpu::tm::station_count(type[t]);
        for (int q = 0; q < ns; ++q) {
            if (stn3[3 * (size_t)t + q] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "block_set_terrestrial: station index out of range");
            for (int r = 0; r < q; ++r)
                if (stn3[3 * (size_t)t + q] == stn3[3 * (size_t)t + r])
                    return fail(ctx, DNAGPU_EINVAL, "block_set_terrestrial: a measurement names one station twice");
        }
        if (!(variance[t] > 0.0)) return fail(ctx, DNAGPU_EINVAL, "block_set_terrestrial: non-positive variance");
        blk0[t] = nb;
        vec0[t] = nv;
        if (type[t] != 'D') nb += (uint32_t)(ns * (ns + 1) / 2);     // (a direction set's blocks: dnagpu_block_set_direction_sets)
        nv += (uint32_t)ns;
    }
    for (void* p : {(void*)b->t_type, (void*)b->t_stn, (void*)b->t_blk0, (void*)b->t_vec0, (void*)b->t_val, (void*)b->t_pre, (void*)b->t_var,
                    (void*)b->t_ih, (void*)b->t_th})
        if (p) hipFree(p);
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c)
        for (void* p : {(void*)b->tb[c], (void*)b->trow[c]})
            if (p) hipFree(p);
    b->t_type = nullptr;
    b->t_stn = b->t_blk0 = b->t_vec0 = nullptr;
    b->t_val = b->t_pre = b->t_var = b->t_ih = b->t_th = nullptr;
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) b->tb[c] = b->trow[c] = nullptr;
    b->n_t = n_t;
    b->n_tblk = nb;
    b->n_tvec = nv;
    b->n_dsblk = 0;
    b->h_ds_ents.clear();
    for (void* p : {(void*)b->ds_a, (void*)b->ds_b, (void*)b->ds_pq, (void*)b->ds_w, (void*)b->ds_row0, (void*)b->ds_k, (void*)b->ds_woff, (void*)b->ds_wts})
        if (p) hipFree(p);
    b->ds_a = b->ds_b = b->ds_pq = b->ds_w = b->ds_row0 = b->ds_k = b->ds_woff = nullptr;
    b->ds_wts = nullptr;
    b->h_ttype.assign(type, type + n_t);
    b->h_tstn.assign(stn3, stn3 + 3 * (size_t)n_t);
    b->h_tpos.assign(cml_pos, cml_pos + n_t);
    b->h_cpos.clear();
    if (cluster_cml_pos) b->h_cpos.assign(cluster_cml_pos, cluster_cml_pos + n_clusters);
    // the W b vectors of the GNSS measurements are followed by the terrestrial ones
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        if (b->wb[c] && b->wb_own) hipFree(b->wb[c]);      // (the arena's vectors are not freed one by one)
        b->wb[c] = nullptr;
        HIPCHK(dnagpu::poison_malloc(&b->wb[c], std::max<size_t>((size_t)b->n_bl + nv, 1) * 3 * sizeof(double)));
    }
    b->wb_own = true;
    if (!n_t) return DNAGPU_OK;
    auto upv = [&](void** dev, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = dnagpu::poison_malloc(dev, bytes);
        if (e == hipSuccess) e = hipMemcpy(*dev, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIPCHK(upv((void**)&b->t_type, type, n_t));
    HIPCHK(upv((void**)&b->t_stn, stn3, 3 * (size_t)n_t * sizeof(uint32_t)));
    HIPCHK(upv((void**)&b->t_blk0, blk0.data(), n_t * sizeof(uint32_t)));
    HIPCHK(upv((void**)&b->t_vec0, vec0.data(), n_t * sizeof(uint32_t)));
    HIPCHK(upv((void**)&b->t_val, value, n_t * sizeof(double)));
    HIPCHK(upv((void**)&b->t_pre, pre_adj_meas, n_t * sizeof(double)));
    HIPCHK(upv((void**)&b->t_var, variance, n_t * sizeof(double)));
    HIPCHK(upv((void**)&b->t_ih, inst_height, n_t * sizeof(double)));
    HIPCHK(upv((void**)&b->t_th, targ_height, n_t * sizeof(double)));
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        HIPCHK(dnagpu::poison_malloc(&b->tb[c], n_t * sizeof(double)));
        HIPCHK(dnagpu::poison_malloc(&b->trow[c], 9 * (size_t)n_t * sizeof(double)));
    }
    return DNAGPU_OK;
}

int dnagpu_block_set_direction_sets(dnagpu_ctx* ctx, uint32_t blk, uint32_t n_sets, const uint32_t* set_first, const uint32_t* set_size,
                                    const double* weights) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b || (n_sets && (!set_first || !set_size || !weights))) return fail(ctx, DNAGPU_EINVAL, "block_set_direction_sets: bad arguments");
    const uint32_t n_t = b->n_t, ns = b->n_stn;
    std::vector<uint32_t> row0(n_t, 0), kk(n_t, 0), woff(n_t, 0), ea, eb, epq, ew;
    std::vector<uint8_t> in_set(n_t, 0);
    b->h_ds_ents.clear();
    size_t wtot = 0;
    for (uint32_t s = 0; s < n_sets; ++s) {
        if (!set_size[s] || (uint64_t)set_first[s] + set_size[s] > n_t) return fail(ctx, DNAGPU_EINVAL, "block_set_direction_sets: bad set range");
        const uint32_t r0 = set_first[s], k = set_size[s];
        for (uint32_t a = 0; a < k; ++a) {
            if (b->h_ttype[r0 + a] != 'D' || in_set[r0 + a] || b->h_tpos[r0 + a] != b->h_tpos[r0])
                return fail(ctx, DNAGPU_EINVAL, "block_set_direction_sets: the rows of a set are consecutive type D entries of one measurement");
            in_set[r0 + a] = 1;
            row0[r0 + a] = r0;
            kk[r0 + a] = k;
            woff[r0 + a] = (uint32_t)wtot;
        }
        // every ordered pair of station slots (row a, p), (row b, q) whose stations satisfy stn(a,p) >= stn(b,q): one block
        for (uint32_t a = 0; a < k; ++a)
            for (int p = 0; p < 3; ++p)
                for (uint32_t c = 0; c < k; ++c)
                    for (int q = 0; q < 3; ++q) {
                        const uint32_t sa = b->h_tstn[3 * (size_t)(r0 + a) + p], sb = b->h_tstn[3 * (size_t)(r0 + c) + q];
                        if (sa < sb) continue;
                        b->h_ds_ents.push_back({(uint64_t)sa * ns + sb, b->h_tpos[r0], (uint32_t)ea.size()});
                        ea.push_back(r0 + a);
                        eb.push_back(r0 + c);
                        epq.push_back((uint32_t)p | ((uint32_t)q << 2));
                        ew.push_back((uint32_t)(wtot + a + (size_t)c * k));
                    }
        wtot += (size_t)k * k;
    }
    for (uint32_t t = 0; t < n_t; ++t)
        if (b->h_ttype[t] == 'D' && !in_set[t]) return fail(ctx, DNAGPU_EINVAL, "block_set_direction_sets: a type D entry belongs to no set");
    auto up32 = [&](uint32_t** dev, const std::vector<uint32_t>& v) -> hipError_t {
        hipError_t e = dnagpu::poison_malloc((void**)dev, std::max<size_t>(v.size(), 1) * sizeof(uint32_t));
        if (e == hipSuccess && !v.empty()) e = hipMemcpy(*dev, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        return e;
    };
    for (void* p : {(void*)b->ds_a, (void*)b->ds_b, (void*)b->ds_pq, (void*)b->ds_w, (void*)b->ds_row0, (void*)b->ds_k, (void*)b->ds_woff, (void*)b->ds_wts})
        if (p) hipFree(p);
    b->ds_a = b->ds_b = b->ds_pq = b->ds_w = b->ds_row0 = b->ds_k = b->ds_woff = nullptr;
    b->ds_wts = nullptr;
    HIPCHK(up32(&b->ds_a, ea));
    HIPCHK(up32(&b->ds_b, eb));
    HIPCHK(up32(&b->ds_pq, epq));
    HIPCHK(up32(&b->ds_w, ew));
    HIPCHK(up32(&b->ds_row0, row0));
    HIPCHK(up32(&b->ds_k, kk));
    HIPCHK(up32(&b->ds_woff, woff));
    HIPCHK(dnagpu::poison_malloc(&b->ds_wts, std::max<size_t>(wtot, 1) * sizeof(double)));
    if (wtot) HIPCHK(hipMemcpy(b->ds_wts, weights, wtot * sizeof(double), hipMemcpyHostToDevice));
    b->n_dsblk = (uint32_t)ea.size();
    return DNAGPU_OK;
}

int dnagpu_block_set_clusters(dnagpu_ctx* ctx, uint32_t blk, const uint32_t* stn1, const uint32_t* stn2, const double* obs,
                              uint32_t n_clusters, const uint32_t* cluster_off, const double* vcv) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "block_set_clusters: unknown block");
    const uint32_t m = b->n_bl, ns = b->n_stn;
    if (m && (!stn1 || !stn2 || !obs || !vcv || !cluster_off)) return fail(ctx, DNAGPU_EINVAL, "block_set_clusters: null argument");
    if (n_clusters && (cluster_off[0] != 0 || cluster_off[n_clusters] != m))
        return fail(ctx, DNAGPU_EINVAL, "block_set_clusters: cluster offsets do not cover the vectors");
    for (uint32_t i = 0; i < m; ++i)
        if ((stn1[i] != DNAGPU_NO_STATION && stn1[i] >= ns) || stn2[i] >= ns)
            return fail(ctx, DNAGPU_EINVAL, "block_set_clusters: station index out of range");

    // weight-block bookkeeping: cluster c (k vectors) owns k*k consecutive 3x3 blocks
    std::vector<uint32_t> wrow(m), c0v(m), kv(m), woff(n_clusters + 1, 0);
    std::vector<size_t> voff(n_clusters + 1, 0);   // offsets of the clusters' variance matrices in `vcv`
    uint32_t kmax = 0;
    for (uint32_t c = 0; c < n_clusters; ++c) {
        if (cluster_off[c + 1] <= cluster_off[c]) return fail(ctx, DNAGPU_EINVAL, "block_set_clusters: empty cluster");
        uint32_t k = cluster_off[c + 1] - cluster_off[c];
        woff[c + 1] = woff[c] + k * k;
        voff[c + 1] = voff[c] + (size_t)9 * k * k;
        kmax = std::max(kmax, k);
        for (uint32_t j = 0; j < k; ++j) {
            uint32_t v = cluster_off[c] + j;
            wrow[v] = woff[c] + j * k;
            c0v[v] = cluster_off[c];
            kv[v] = k;
        }
    }
    const uint32_t n_wblk = woff[n_clusters];

    // station-pair structure: contributions sorted by (row, col), CML order inside a pair.  A cluster contributes,
    // for every ordered pair of its vectors (j, j') and every pair of their end stations (a, a') with a >= a',
    // sign(a) sign(a') W(j, j'); the mirrored (a < a') case is produced by the pair (j', j).
    struct Ent {
        uint64_t key;
        uint32_t pos;   // position of the measurement in the block's CML (contributions are summed in that order)
        uint32_t ent;
    };
    std::vector<Ent> ents;
    ents.reserve((size_t)m * 3 + (size_t)b->n_t * 3);
    const bool have_cpos = b->h_cpos.size() == n_clusters && n_clusters > 0;
    auto cluster_pos = [&](uint32_t c) { return have_cpos ? b->h_cpos[c] : c; };
    for (uint32_t c = 0; c < n_clusters; ++c) {
        const uint32_t v0 = cluster_off[c], k = cluster_off[c + 1] - v0;
        for (uint32_t j = 0; j < k; ++j)
            for (uint32_t jp = 0; jp < k; ++jp) {
                const uint32_t blkidx = woff[c] + j * k + jp;
                // end stations: second station (+1) first, then the first station (-1): for a single baseline this is
                // the order of UpdateNormals_G (dnaadjust.cpp:1664-1684)
                const uint32_t ea[2] = {stn2[v0 + j], stn1[v0 + j]};
                const uint32_t eb[2] = {stn2[v0 + jp], stn1[v0 + jp]};
                for (int x = 0; x < 2; ++x) {
                    if (ea[x] == DNAGPU_NO_STATION) continue;
                    for (int y = 0; y < 2; ++y) {
                        if (eb[y] == DNAGPU_NO_STATION) continue;
                        if (ea[x] < eb[y]) continue;
                        ents.push_back({(uint64_t)ea[x] * ns + eb[y], cluster_pos(c), (blkidx << 1) | (uint32_t)(x != y)});
                    }
                }
            }
    }
    // terrestrial measurements: one 3x3 block w a_p^T a_q per station pair with local(p) >= local(q), enumerated as
    // tmsr_eval_kernel writes them; block indices continue after the GNSS weight blocks
    for (uint32_t t = 0, tb = 0; t < b->n_t; ++t) {
        if (b->h_ttype[t] == 'D') {
            if (b->n_dsblk == 0 && b->h_ds_ents.empty()) return fail(ctx, DNAGPU_EINVAL, "block_set_clusters: type D entries without dnagpu_block_set_direction_sets");
            continue;
        }
        const int nst = dnagpu::tm::station_count((char)b->h_ttype[t]);
        const uint32_t* l = &b->h_tstn[3 * (size_t)t];
        for (int pp = 0; pp < nst; ++pp)
            for (int qq = 0; qq < nst; ++qq) {
                if (pp != qq && !(l[pp] > l[qq])) continue;
                ents.push_back({(uint64_t)l[pp] * ns + l[qq], b->h_tpos[t], ((n_wblk + tb) << 1)});
                ++tb;
            }
    }
    // direction sets: their blocks sit behind the per-measurement terrestrial ones
    for (const auto& d : b->h_ds_ents) ents.push_back({d.key, d.pos, ((n_wblk + b->n_tblk + d.blk) << 1)});
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.key != y.key ? x.key < y.key : x.pos < y.pos; });
    std::vector<uint32_t> prow, pcol, poff, pent(ents.size());
    for (size_t k = 0; k < ents.size(); ++k) {
        if (k == 0 || ents[k].key != ents[k - 1].key) {
            prow.push_back((uint32_t)(ents[k].key / ns));
            pcol.push_back((uint32_t)(ents[k].key % ns));
            poff.push_back((uint32_t)k);
        }
        pent[k] = ents[k].ent;
    }
    poff.push_back((uint32_t)ents.size());
    // incidence per station (CML order): vector * 2 + (1 if the station is the vector's second / only station)
    // (terrestrial measurements add "virtual" vectors a_p w b behind the GNSS ones, always with a positive sign)
    struct Inc {
        uint32_t stn, pos, ent;
    };
    std::vector<Inc> incs;
    incs.reserve((size_t)m * 2 + b->n_tvec);
    for (uint32_t c = 0; c < n_clusters; ++c)
        for (uint32_t i = cluster_off[c]; i < cluster_off[c + 1]; ++i) {
            if (stn1[i] != DNAGPU_NO_STATION) incs.push_back({stn1[i], cluster_pos(c), i * 2u});
            incs.push_back({stn2[i], cluster_pos(c), i * 2u + 1u});
        }
    for (uint32_t t = 0, tv = 0; t < b->n_t; ++t) {
        const int nst = dnagpu::tm::station_count((char)b->h_ttype[t]);
        for (int q = 0; q < nst; ++q, ++tv) incs.push_back({b->h_tstn[3 * (size_t)t + q], b->h_tpos[t], (m + tv) * 2u + 1u});
    }
    std::stable_sort(incs.begin(), incs.end(), [](const Inc& x, const Inc& y) { return x.stn != y.stn ? x.stn < y.stn : x.pos < y.pos; });
    std::vector<uint32_t> ioff(ns + 1, 0), inc(incs.size());
    for (size_t k = 0; k < incs.size(); ++k) {
        ioff[incs[k].stn + 1]++;
        inc[k] = incs[k].ent;
    }
    for (uint32_t s = 0; s < ns; ++s) ioff[s + 1] += ioff[s];

    for (void* p : {(void*)b->pair_row, (void*)b->pair_col, (void*)b->pair_off, (void*)b->pair_ent, (void*)b->inc_off, (void*)b->inc,
                    (void*)b->Wblk})
        if (p) hipFree(p);
    b->pair_row = b->pair_col = b->pair_off = b->pair_ent = b->inc_off = b->inc = nullptr;
    b->Wblk = nullptr;
    b->n_pairs = (uint32_t)prow.size();
    b->n_wblk = n_wblk;
    auto up = [&](uint32_t** dev, const std::vector<uint32_t>& v) -> hipError_t {
        size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(uint32_t);
        hipError_t e = dnagpu::poison_malloc((void**)dev, bytes);
        if (e == hipSuccess && !v.empty()) e = hipMemcpy(*dev, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        return e;
    };
    HIPCHK(up(&b->pair_row, prow));
    HIPCHK(up(&b->pair_col, pcol));
    HIPCHK(up(&b->pair_off, poff));
    HIPCHK(up(&b->pair_ent, pent));
    HIPCHK(up(&b->inc_off, ioff));
    HIPCHK(up(&b->inc, inc));
    HIPCHK(dnagpu::poison_malloc(&b->Wblk, std::max<size_t>((size_t)n_wblk + (size_t)DNAGPU_NUM_CHAINS * ((size_t)b->n_tblk + b->n_dsblk), 1) * 9 * sizeof(double)));
    if (!m) return DNAGPU_OK;
    HIPCHK(hipMemcpy(b->s1, stn1, (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->s2, stn2, (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->obs, obs, (size_t)m * 3 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->vec_wrow, wrow.data(), (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->vec_c0, c0v.data(), (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->vec_k, kv.data(), (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice));

    // W = V^-1 per cluster (LoadVarianceMatrix_G/X/Y -> FormInverseVarianceMatrix, dnaadjust.cpp:4214/4312/4494, 8472).
    // single vectors: one thread each, registers only; larger clusters: the dense inverse of the device layer
    std::vector<double> v6;
    std::vector<uint32_t> dst;
    for (uint32_t c = 0; c < n_clusters; ++c)
        if (cluster_off[c + 1] - cluster_off[c] == 1) {
            const double* V = vcv + voff[c];   // 3x3 column-major
            const double six[6] = {V[0], V[3], V[4], V[6], V[7], V[8]};   // xx, xy, yy, xz, yz, zz (upper triangle)
            v6.insert(v6.end(), six, six + 6);
            dst.push_back(woff[c]);
        }
    int bad = 0x7fffffff;
    if (!dst.empty()) {
        double* vtmp = nullptr;
        uint32_t* dtmp = nullptr;
        HIPCHK(dnagpu::poison_malloc(&vtmp, v6.size() * sizeof(double)));
        hipError_t e = dnagpu::poison_malloc(&dtmp, dst.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(vtmp, v6.data(), v6.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dtmp, dst.data(), dst.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ctx->bad_dev, &bad, sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_weights(vtmp, dtmp, b->Wblk, (uint32_t)dst.size(), ctx->bad_dev, ctx->stream[0]);
            e = hipStreamSynchronize(ctx->stream[0]);
        }
        if (e == hipSuccess) e = hipMemcpy(&bad, ctx->bad_dev, sizeof(int), hipMemcpyDeviceToHost);
        hipFree(vtmp);
        if (dtmp) hipFree(dtmp);
        if (e != hipSuccess) return fail(ctx, DNAGPU_EHIP, "block_set_clusters: weights", e);
        if (bad != 0x7fffffff) {
            char buf[160];
            snprintf(buf, sizeof(buf), "Matrix inversion failed, the matrix is singular. (variance matrix of measurement %d)", bad);
            note_error(ctx, buf, bad + 1);
            return DNAGPU_ENOTPOSDEF;
        }
    }
    if (kmax > 1) {
        dnagpu_matrix* vm = nullptr;
        int rc = dnagpu_matrix_create(ctx, 3 * kmax, &vm);
        if (rc) return rc;
        for (uint32_t c = 0; c < n_clusters && !rc; ++c) {
            const uint32_t k = cluster_off[c + 1] - cluster_off[c];
            if (k == 1) continue;
            const uint32_t n = 3 * k;
            rc = dnagpu_matrix_reset(ctx, 0, vm, n);
            if (!rc) {
                hipError_t e = hipMemcpy2DAsync(vm->F, (size_t)vm->np * sizeof(double), vcv + voff[c], (size_t)n * sizeof(double),
                                                (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, ctx->stream[0]);
                if (e != hipSuccess) rc = fail(ctx, DNAGPU_EHIP, "block_set_clusters: variance upload", e);
            }
            if (!rc) rc = dnagpu_invert(ctx, 0, vm, 0);
            if (!rc) {
                launch_cluster_blocks(vm->F, vm->np, k, b->Wblk + (size_t)woff[c] * 9, ctx->stream[0]);
                if (hipStreamSynchronize(ctx->stream[0]) != hipSuccess) rc = fail(ctx, DNAGPU_EHIP, "block_set_clusters: weight blocks");
            }
        }
        dnagpu_matrix_destroy(ctx, vm);
        if (rc) return rc;
    }
    return DNAGPU_OK;
}

int dnagpu_block_set_baselines(dnagpu_ctx* ctx, uint32_t blk, const uint32_t* stn1, const uint32_t* stn2, const double* obs,
                               const double* vcv6) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "block_set_baselines: unknown block");
    const uint32_t m = b->n_bl;
    if (m && (!stn1 || !stn2 || !obs || !vcv6)) return fail(ctx, DNAGPU_EINVAL, "block_set_baselines: null argument");
    for (uint32_t i = 0; i < m; ++i)
        if (stn1[i] == DNAGPU_NO_STATION) return fail(ctx, DNAGPU_EINVAL, "block_set_baselines: station index out of range");
    // every baseline is a cluster of one vector
    std::vector<uint32_t> off(m + 1);
    std::vector<double> vcv((size_t)m * 9);
    for (uint32_t i = 0; i <= m; ++i) off[i] = i;
    for (uint32_t i = 0; i < m; ++i) {
        const double* v = vcv6 + (size_t)i * 6;
        double* V = vcv.data() + (size_t)i * 9;
        V[0] = v[0]; V[1] = v[1]; V[2] = v[3];
        V[3] = v[1]; V[4] = v[2]; V[5] = v[4];
        V[6] = v[3]; V[7] = v[4]; V[8] = v[5];
    }
    return dnagpu_block_set_clusters(ctx, blk, stn1, stn2, obs, m, off.data(), vcv.data());
}

int dnagpu_block_get_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, int which, double* xyz) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !station_vec(*b, which, chain) || (!xyz && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_get_stations: bad arguments");
    if (!b->n_stn) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(xyz, station_vec(*b, which, chain), 3 * (size_t)b->n_stn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_block_put_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, int which, const double* xyz) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !station_vec(*b, which, chain) || (!xyz && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_put_stations: bad arguments");
    if (!b->n_stn) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(station_vec(*b, which, chain), xyz, 3 * (size_t)b->n_stn * sizeof(double), hipMemcpyHostToDevice, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_block_copy_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, int dst_which, int src_which) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !station_vec(*b, dst_which, chain) || !station_vec(*b, src_which, chain))
        return fail(ctx, DNAGPU_EINVAL, "block_copy_stations: bad arguments");
    if (!b->n_stn || dst_which == src_which) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(station_vec(*b, dst_which, chain), station_vec(*b, src_which, chain), 3 * (size_t)b->n_stn * sizeof(double),
                          hipMemcpyDeviceToDevice, ctx->stream[chain]));
    return DNAGPU_OK;
}

static int d2h_early(dnagpu_ctx* ctx, int chain, void* dst, const void* src, size_t bytes) {
    if (!bytes) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_block_compute_b(dnagpu_ctx* ctx, int chain, uint32_t blk) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "block_compute_b: unknown block");
    launch_compute_b(b->s1, b->s2, b->obs, b->x_est[chain], b->b[chain], b->n_bl, ctx->stream[chain]);
    if (b->n_t) {
        if (!b->s_llh || !b->Wblk) return fail(ctx, DNAGPU_EINVAL, "block_compute_b: station records / measurement lists not set");
        launch_tmsr_eval(b->t_type, b->t_stn, b->t_val, b->t_pre, b->t_var, b->t_ih, b->t_th, b->t_blk0, b->t_vec0, b->x_est[chain], b->s_llh,
                         b->s_geoid, b->s_defl, b->tb[chain], b->trow[chain],
                         b->Wblk + ((size_t)b->n_wblk + (size_t)chain * ((size_t)b->n_tblk + b->n_dsblk)) * 9, b->wb[chain], b->n_bl, b->n_t,
                         ctx->stream[chain]);
        if (b->ds_k)
            launch_dsets(b->ds_a, b->ds_b, b->ds_pq, b->ds_w, b->ds_wts, b->trow[chain],
                         b->Wblk + ((size_t)b->n_wblk + (size_t)chain * ((size_t)b->n_tblk + b->n_dsblk) + b->n_tblk) * 9, b->n_dsblk, b->ds_row0,
                         b->ds_k, b->ds_woff, b->tb[chain], b->t_vec0, b->wb[chain], b->n_bl, b->n_t, ctx->stream[chain]);
    }
    return DNAGPU_OK;
}

int dnagpu_block_update_geodetic(dnagpu_ctx* ctx, int chain, uint32_t blk) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !b->s_llh) return fail(ctx, DNAGPU_EINVAL, "block_update_geodetic: unknown block or no station records");
    launch_geodetic(b->x_est[chain], b->s_llh, b->n_stn, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_block_get_station_llh(dnagpu_ctx* ctx, int chain, uint32_t blk, double* llh) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !b->s_llh || (!llh && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_get_station_llh: bad arguments");
    return d2h_early(ctx, chain, llh, b->s_llh, 3 * (size_t)b->n_stn * sizeof(double));
}

int dnagpu_block_get_terrestrial(dnagpu_ctx* ctx, int chain, uint32_t blk, double* meas_minus_comp, double* design_rows) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "block_get_terrestrial: unknown block");
    if (!b->n_t) return DNAGPU_OK;
    int rc = DNAGPU_OK;
    if (meas_minus_comp) rc = d2h_early(ctx, chain, meas_minus_comp, b->tb[chain], (size_t)b->n_t * sizeof(double));
    if (!rc && design_rows) rc = d2h_early(ctx, chain, design_rows, b->trow[chain], 9 * (size_t)b->n_t * sizeof(double));
    return rc;
}

int dnagpu_block_terrestrial_precisions(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_matrix* variances, double* prec) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !variances || variances->n != 3 * b->n_stn || (b->n_t && !prec))
        return fail(ctx, DNAGPU_EINVAL, "block_terrestrial_precisions: bad arguments");
    if (!b->n_t) return DNAGPU_OK;
    int rc = ensure_scr_f64(ctx, chain, b->n_t);
    if (rc) return rc;
    launch_tmsr_stats(b->t_type, b->t_stn, b->trow[chain], variances->F, variances->np, ctx->scr_f64[chain], b->n_t, ctx->stream[chain]);
    return d2h_early(ctx, chain, prec, ctx->scr_f64[chain], (size_t)b->n_t * sizeof(double));
}

static int d2h(dnagpu_ctx* ctx, int chain, void* dst, const void* src, size_t bytes) {
    if (!bytes) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_block_get_b(dnagpu_ctx* ctx, int chain, uint32_t blk, double* out) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || (!out && b->n_bl)) return fail(ctx, DNAGPU_EINVAL, "block_get_b: bad arguments");
    return d2h(ctx, chain, out, b->b[chain], (size_t)b->n_bl * 3 * sizeof(double));
}

int dnagpu_block_get_weights(dnagpu_ctx* ctx, int chain, uint32_t blk, double* w6) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || (!w6 && b->n_bl)) return fail(ctx, DNAGPU_EINVAL, "block_get_weights: bad arguments");
    if (!b->n_bl) return DNAGPU_OK;
    int rc = ensure_scr_f64(ctx, chain, (size_t)b->n_bl * 6);
    if (rc) return rc;
    launch_diag_weights(b->Wblk, b->vec_wrow, b->vec_c0, ctx->scr_f64[chain], b->n_bl, ctx->stream[chain]);
    return d2h(ctx, chain, w6, ctx->scr_f64[chain], (size_t)b->n_bl * 6 * sizeof(double));
}

int dnagpu_block_msr_statistics(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_matrix* variances, double* prec6, double* chi) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || (b->n_bl && !chi) || (variances && (variances->n != 3 * b->n_stn || !prec6)))
        return fail(ctx, DNAGPU_EINVAL, "block_msr_statistics: bad arguments");
    if (!b->n_bl) return DNAGPU_OK;
    int rc = ensure_scr_f64(ctx, chain, (size_t)b->n_bl * 7);
    if (rc) return rc;
    double* dprec = ctx->scr_f64[chain];
    double* dchi = dprec + (size_t)b->n_bl * 6;
    launch_msr_stats(b->Wblk, b->vec_wrow, b->vec_c0, b->vec_k, b->s1, b->s2, b->b[chain], b->wb[chain], variances ? variances->F : nullptr,
                     variances ? variances->np : 0, dprec, dchi, b->n_bl, ctx->stream[chain]);
    if (variances) {
        HIPCHK(hipMemcpyAsync(prec6, dprec, (size_t)b->n_bl * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream[chain]));
    }
    return d2h(ctx, chain, chi, dchi, (size_t)b->n_bl * sizeof(double));
}

int dnagpu_block_get_corrections(dnagpu_ctx* ctx, int chain, uint32_t blk, double* corr) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (chain < 0) {         // the corrections set aside by dnagpu_block_keep_corrections
        if (!b || !b->corr_keep || (!corr && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_get_corrections: nothing kept");
        return d2h(ctx, 0, corr, b->corr_keep, (size_t)b->n_stn * 3 * sizeof(double));
    }
    CHK_CHAIN();
    if (!b || (!corr && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_get_corrections: bad arguments");
    return d2h(ctx, chain, corr, b->corr[chain], (size_t)b->n_stn * 3 * sizeof(double));
}

/* ---- oscillation diagnostics (dna_adjust::UpdateIterationDiagnostics, ADJ:7450-7554) ---------------------------------------- */
int dnagpu_osc_reset(dnagpu_ctx* ctx, size_t n_stations) {
    CHK_CTX();
    if (ctx->osc_stations != n_stations) {
        HIPCHK(hipDeviceSynchronize());
        for (void* p : {(void*)ctx->osc_prev, (void*)ctx->osc_seen, (void*)ctx->osc_cnt})
            if (p) hipFree(p);
        ctx->osc_prev = nullptr;
        ctx->osc_seen = ctx->osc_cnt = nullptr;
        ctx->osc_stations = 0;
        ctx->osc_key = 0;       // (the visit lists of dnagpu_osc_blocks are per station of the network)
        if (n_stations) {
            hipError_t e = dnagpu::poison_malloc(&ctx->osc_prev, 3 * n_stations * sizeof(double));
            if (e == hipSuccess) e = dnagpu::poison_malloc(&ctx->osc_seen, n_stations * sizeof(uint32_t));
            if (e == hipSuccess) e = dnagpu::poison_malloc(&ctx->osc_cnt, n_stations * sizeof(uint32_t));
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, DNAGPU_ENOMEM, "oscillation diagnostics", e);
            }
            ctx->osc_stations = n_stations;
        }
    }
    if (!ctx->osc_flagged) HIPCHK(dnagpu::poison_malloc(&ctx->osc_flagged, sizeof(uint32_t)));
    hipStream_t st = ctx->stream[0];
    if (n_stations) {
        HIPCHK(hipMemsetAsync(ctx->osc_seen, 0, n_stations * sizeof(uint32_t), st));
        HIPCHK(hipMemsetAsync(ctx->osc_cnt, 0, n_stations * sizeof(uint32_t), st));
    }
    HIPCHK(hipMemsetAsync(ctx->osc_flagged, 0, sizeof(uint32_t), st));
    return DNAGPU_OK;
}

int dnagpu_block_keep_corrections(dnagpu_ctx* ctx, int chain, uint32_t blk) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !b->corr[chain]) return fail(ctx, DNAGPU_EINVAL, "block_keep_corrections: bad arguments");
    if (!b->n_stn) return DNAGPU_OK;
    if (!b->corr_keep) HIPCHK(dnagpu::poison_malloc(&b->corr_keep, (size_t)b->n_stn * 3 * sizeof(double)));
    HIPCHK(hipMemcpyAsync(b->corr_keep, b->corr[chain], (size_t)b->n_stn * 3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream[chain]));
    return DNAGPU_OK;
}

int dnagpu_osc_block(dnagpu_ctx* ctx, uint32_t blk, int corr_chain, const uint32_t* stations) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    const double* corr = !b ? nullptr : corr_chain < 0 ? b->corr_keep : corr_chain < DNAGPU_NUM_CHAINS ? b->corr[corr_chain] : nullptr;
    if (!b || !corr || (!stations && b->n_stn) || !ctx->osc_flagged) return fail(ctx, DNAGPU_EINVAL, "osc_block: bad arguments");
    hipStream_t st = ctx->stream[0];
    if (!b->osc_gidx && b->n_stn) {
        for (uint32_t s = 0; s < b->n_stn; ++s)
            if (stations[s] >= ctx->osc_stations) return fail(ctx, DNAGPU_EINVAL, "osc_block: station out of range");
        HIPCHK(dnagpu::poison_malloc(&b->osc_gidx, (size_t)b->n_stn * sizeof(uint32_t)));
        HIPCHK(dnagpu::poison_malloc(&b->osc_visit, (size_t)b->n_stn * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(b->osc_gidx, stations, (size_t)b->n_stn * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    launch_osc_update(corr, b->osc_gidx, b->n_stn, ctx->osc_prev, ctx->osc_seen, ctx->osc_cnt, b->osc_visit, ctx->osc_flagged, st);
    return DNAGPU_OK;
}

int dnagpu_osc_blocks(dnagpu_ctx* ctx, uint32_t n, const uint32_t* blks, const int* corr_chain, const uint32_t* const* stations) {
    CHK_CTX();
    if (!n) return DNAGPU_OK;
    if (!blks || !corr_chain || !stations || !ctx->osc_flagged) return fail(ctx, DNAGPU_EINVAL, "osc_blocks: bad arguments");
    std::vector<OscRow> rows(n);
    for (uint32_t q = 0; q < n; ++q) {
        Block* b = find_block(ctx, blks[q]);
        const double* corr = !b ? nullptr : corr_chain[q] < 0 ? b->corr_keep : corr_chain[q] < DNAGPU_NUM_CHAINS ? b->corr[corr_chain[q]] : nullptr;
        if (!b || !corr || (!stations[q] && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "osc_blocks: bad arguments");
        if (!b->osc_gidx && b->n_stn) {
            for (uint32_t s = 0; s < b->n_stn; ++s)
                if (stations[q][s] >= ctx->osc_stations) return fail(ctx, DNAGPU_EINVAL, "osc_blocks: station out of range");
            HIPCHK(dnagpu::poison_malloc(&b->osc_gidx, (size_t)b->n_stn * sizeof(uint32_t)));
            HIPCHK(dnagpu::poison_malloc(&b->osc_visit, (size_t)b->n_stn * sizeof(uint32_t)));
            HIPCHK(hipMemcpy(b->osc_gidx, stations[q], (size_t)b->n_stn * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        rows[q] = {corr, b->osc_gidx, b->osc_visit, b->n_stn};
    }
    hipStream_t st = ctx->stream[0];
    // the stations' visit lists (block order) and the rows' table: built and uploaded when the set of blocks or their vectors change
    // (the visit lists depend on the blocks only; which chain's corrections a block is read from changes from iteration to iteration -- the
    //  rows' table is uploaded again when it does, 32 bytes per block)
    uint64_t key = 1469598103934665603ull ^ (uint64_t)n;
    for (uint32_t q = 0; q < n; ++q) key = (key ^ blks[q]) * 1099511628211ull;
    // (the key is a hash: the list of block ids itself decides; dnagpu_block_create / _destroy clear the key, a block's station list is
    //  set once, by its first visit here)
    if (key == ctx->osc_key && ctx->osc_rows && ctx->osc_blks.size() == n && !memcmp(ctx->osc_blks.data(), blks, (size_t)n * sizeof(uint32_t))) {
        if (ctx->osc_rows_host.size() != (size_t)n * sizeof(OscRow) || memcmp(ctx->osc_rows_host.data(), rows.data(), (size_t)n * sizeof(OscRow))) {
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipMemcpy(ctx->osc_rows, rows.data(), (size_t)n * sizeof(OscRow), hipMemcpyHostToDevice));
            ctx->osc_rows_host.assign((const uint8_t*)rows.data(), (const uint8_t*)rows.data() + (size_t)n * sizeof(OscRow));
        }
    } else {
        HIPCHK(hipStreamSynchronize(st));
        for (void* p : {ctx->osc_rows, (void*)ctx->osc_off, ctx->osc_visits})
            if (p) hipFree(p);
        ctx->osc_rows = ctx->osc_visits = nullptr;
        ctx->osc_off = nullptr;
        ctx->osc_key = 0;
        std::vector<uint32_t> off(ctx->osc_stations + 1, 0);
        for (uint32_t q = 0; q < n; ++q)
            for (uint32_t s = 0; s < rows[q].n_stn; ++s) off[stations[q][s] + 1]++;
        for (size_t g = 0; g < ctx->osc_stations; ++g) off[g + 1] += off[g];
        std::vector<uint32_t> fill(off.begin(), off.end() - 1);
        std::vector<uint2> visits(off.back() ? off.back() : 1);
        for (uint32_t q = 0; q < n; ++q)          // (blocks in the order given: a station's visits end up in block order)
            for (uint32_t s = 0; s < rows[q].n_stn; ++s) visits[fill[stations[q][s]]++] = make_uint2(q, s);
        HIPCHK(dnagpu::poison_malloc(&ctx->osc_rows, (size_t)n * sizeof(OscRow)));
        HIPCHK(dnagpu::poison_malloc(&ctx->osc_off, off.size() * sizeof(uint32_t)));
        HIPCHK(dnagpu::poison_malloc(&ctx->osc_visits, visits.size() * sizeof(uint2)));
        HIPCHK(hipMemcpy(ctx->osc_rows, rows.data(), (size_t)n * sizeof(OscRow), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->osc_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->osc_visits, visits.data(), visits.size() * sizeof(uint2), hipMemcpyHostToDevice));
        ctx->osc_rows_host.assign((const uint8_t*)rows.data(), (const uint8_t*)rows.data() + (size_t)n * sizeof(OscRow));
        ctx->osc_blks.assign(blks, blks + n);
        ctx->osc_key = key;
    }
    launch_osc_update_stations((const OscRow*)ctx->osc_rows, ctx->osc_off, ctx->osc_visits, (uint32_t)ctx->osc_stations, ctx->osc_prev, ctx->osc_seen,
                               ctx->osc_cnt, ctx->osc_flagged, st);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

int dnagpu_osc_flagged(dnagpu_ctx* ctx, uint32_t* n_flagged) {
    CHK_CTX();
    if (!n_flagged || !ctx->osc_flagged) return fail(ctx, DNAGPU_EINVAL, "osc_flagged: bad arguments");
    hipStream_t st = ctx->stream[0];
    HIPCHK(hipMemcpyAsync(n_flagged, ctx->osc_flagged, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(ctx->osc_flagged, 0, sizeof(uint32_t), st));
    HIPCHK(hipStreamSynchronize(st));
    return DNAGPU_OK;
}

int dnagpu_osc_block_visits(dnagpu_ctx* ctx, uint32_t blk, uint32_t* visit) {
    CHK_CTX();
    Block* b = find_block(ctx, blk);
    if (!b || !b->osc_visit || !visit) return fail(ctx, DNAGPU_EINVAL, "osc_block_visits: bad arguments");
    HIPCHK(hipMemcpyAsync(visit, b->osc_visit, (size_t)b->n_stn * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream[0]));
    HIPCHK(hipStreamSynchronize(ctx->stream[0]));
    return DNAGPU_OK;
}

int dnagpu_block_get_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, double* rhs) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || (!rhs && b->n_stn)) return fail(ctx, DNAGPU_EINVAL, "block_get_rhs: bad arguments");
    return d2h(ctx, chain, rhs, b->rhs[chain], (size_t)b->n_stn * 3 * sizeof(double));
}

int dnagpu_form_normals(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !m || 3 * b->n_stn > m->n_max) return fail(ctx, DNAGPU_EINVAL, "form_normals: bad arguments");
    m->n = 3 * b->n_stn;
    m->np = pad128(m->n);
    launch_init_padded(m->F, m->n, m->np, ctx->stream[chain]);
    launch_form_normals(b->pair_row, b->pair_col, b->pair_off, b->pair_ent, b->Wblk, m->F, m->np, b->n_pairs, b->n_wblk,
                        (uint32_t)chain * (b->n_tblk + b->n_dsblk), ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_add_diag3x3(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const uint32_t* stn, const double* w9, size_t k, int sign) {
    CHK_CTX();
    CHK_CHAIN();
    if (!m || (k && (!stn || !w9))) return fail(ctx, DNAGPU_EINVAL, "add_diag3x3: bad arguments");
    if (!k) return DNAGPU_OK;
    for (size_t i = 0; i < k; ++i)
        if (3 * (uint64_t)stn[i] + 2 >= m->n) return fail(ctx, DNAGPU_EINVAL, "add_diag3x3: station out of range");
    uint32_t* dstn = nullptr;
    double* dw = nullptr;
    int rc = stage_u32(ctx, chain, stn, k, &dstn);
    if (!rc) rc = stage_f64(ctx, chain, w9, k * 9, &dw);
    if (rc) return rc;
    launch_add_diag3x3(m->F, m->np, dstn, dw, (uint32_t)k, sign < 0 ? -1.0 : 1.0, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_form_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "form_rhs: unknown block");
    launch_form_rhs(b->Wblk, b->vec_wrow, b->vec_c0, b->vec_k, b->b[chain], b->wb[chain], b->n_bl, b->inc_off, b->inc, b->rhs[chain], b->n_stn,
                    ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_form_rhs_batched(dnagpu_ctx* ctx, int chain, int nb, const uint32_t* blks) {
    CHK_CTX();
    CHK_CHAIN();
    if (nb < 1 || nb > BLOCK_BATCH_MAX || !blks) return fail(ctx, DNAGPU_EINVAL, "form_rhs_batched: bad arguments");
    RhsBatch rb{};
    for (int q = 0; q < nb; ++q) {
        Block* b = find_block(ctx, blks[q]);
        if (!b) return fail(ctx, DNAGPU_EINVAL, "form_rhs_batched: unknown block");
        rb.m[q] = {b->Wblk, b->vec_wrow, b->vec_c0, b->vec_k, b->b[chain], b->wb[chain], b->inc_off, b->inc, b->rhs[chain], b->n_bl, b->n_stn};
    }
    gemm_profile_close(ctx->ws[chain]);
    launch_form_rhs_batch(rb, nb, ctx->stream[chain]);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

int dnagpu_solve_corrections(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_matrix* m) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !m || m->n != 3 * b->n_stn) return fail(ctx, DNAGPU_EINVAL, "solve_corrections: bad arguments");
    if (!m->n) return DNAGPU_OK;
    int rc = ensure_symv(ctx, chain, m->np);
    if (rc) return rc;
    HbmTimed timed(ctx, chain, DNAGPU_HBM_SYMV, 8.0 * (double)m->n * m->np);
    launch_symv(m->F, b->rhs[chain], b->corr[chain], ctx->symv_part[chain], m->n, m->np, SYMV_CHUNKS, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_update_estimates(dnagpu_ctx* ctx, int chain, uint32_t blk, double* max_corr, uint32_t* max_row) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b) return fail(ctx, DNAGPU_EINVAL, "update_estimates: unknown block");
    uint32_t n = 3 * b->n_stn;
    double* dval = b->red[chain];
    uint32_t* didx = reinterpret_cast<uint32_t*>(b->red[chain] + 1);
    launch_update_estimates(b->x_est[chain], b->corr[chain], n, dval, didx, ctx->stream[chain]);
    HIPCHK(hipMemcpyAsync(ctx->red_val_host[chain], dval, sizeof(double), hipMemcpyDeviceToHost, ctx->stream[chain]));
    HIPCHK(hipMemcpyAsync(ctx->red_idx_host[chain], didx, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    if (max_corr) *max_corr = *ctx->red_val_host[chain];
    if (max_row) *max_row = *ctx->red_idx_host[chain];
    return DNAGPU_OK;
}

/* ---- junction carry ------------------------------------------------------------ */
int dnagpu_junction_gather(dnagpu_ctx* ctx, int chain, uint32_t blk_from, const dnagpu_matrix* src, const uint32_t* idx_from, size_t k,
                           dnagpu_matrix* jm) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk_from);
    if (!b || !jm || (k && !idx_from) || 3 * k > jm->n_max || (src && src->n != 3 * b->n_stn) || (!src && jm->n != 3 * k))
        return fail(ctx, DNAGPU_EINVAL, "junction_gather: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (idx_from[i] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "junction_gather: station out of range");
    if (src) {
        jm->n = (uint32_t)(3 * k);
        jm->np = pad128(jm->n);
        launch_init_padded(jm->F, jm->n, jm->np, ctx->stream[chain]);
    }
    if (!k) return DNAGPU_OK;
    uint32_t* didx = nullptr;
    int rc = stage_u32(ctx, chain, idx_from, k, &didx);
    if (rc) return rc;
    if (src) launch_junction_gather(src->F, src->np, didx, (uint32_t)k, jm->F, jm->np, ctx->stream[chain]);
    launch_gather_vec3(b->x_est[chain], didx, (uint32_t)k, jm->jest, ctx->stream[chain]);
    jm->form = 0;
    return DNAGPU_OK;
}

}  // extern "C"

namespace {
// The elimination behind dnagpu_schur_carry / dnagpu_block_reduce: the unknowns of the k listed stations are moved behind
// all others (plus one row that carries the right-hand side) and the others are eliminated.  On return (stream ordered)
// T points at the trailing block inside the chain's W workspace: rows / columns 0..3k-1 hold the Schur complement (lower),
// row 3k the reduced right-hand side; ldt its leading dimension.  m (the normals) is destroyed.
// `form` (with `keep`, m = nullptr): the normals are formed here, directly in the elimination's order, with these constraints added
struct FormInOrder {
    const uint32_t* con_stn;      // host
    const double* con_w9;         // host
    size_t n_con;
};
// The unknown order of an elimination onto the k listed stations -- device lists, cached per block (two slots: a block's forward and
// reverse lists alternate) -- for a matrix of nip + (nj + 1 padded) = npp rows
int schur_order(dnagpu_ctx* ctx, Block* b, const uint32_t* idx_out, size_t k, uint32_t nip, uint32_t nj, uint32_t npp, int* slot_out,
                const int32_t** map_dev, const uint32_t** spos_dev) {
    // unknown order: the other stations (block order), padding, the listed stations (list order), the rhs row, padding
    int slot = -1;
    std::unique_lock<std::mutex> lk(ctx->schur_mutex);
    for (int q = 0; q < 2; ++q)
        if (b->schur_map[q] && b->h_schur_nip[q] == nip && b->h_schur_npp[q] == npp && b->h_schur_idx[q].size() == k &&
            std::equal(idx_out, idx_out + k, b->h_schur_idx[q].begin()))
            slot = q;
    if (slot < 0) {
        std::vector<uint8_t> out(b->n_stn, 0);
        for (size_t i = 0; i < k; ++i) {
            if (idx_out[i] >= b->n_stn || out[idx_out[i]]) return fail(ctx, DNAGPU_EINVAL, "schur: bad station list");
            out[idx_out[i]] = 1;
        }
        std::vector<int32_t> map(npp, -1);
        std::vector<uint32_t> spos(b->n_stn, 0);
        uint32_t pos = 0;
        for (uint32_t s = 0; s < b->n_stn; ++s)
            if (!out[s]) {
                spos[s] = pos;
                for (int c = 0; c < 3; ++c) map[pos++] = (int32_t)(3 * s + c);
            }
        for (size_t i = 0; i < k; ++i) {
            spos[idx_out[i]] = nip + 3 * (uint32_t)i;
            for (int c = 0; c < 3; ++c) map[nip + 3 * i + c] = (int32_t)(3 * idx_out[i] + c);
        }
        map[nip + nj] = -2;
        slot = b->schur_map[0] ? 1 : 0;
        // another chain's thread may hold the replaced slot's pointers for a launch it has not enqueued yet: the old lists are
        // retired, not freed at once.  A block that keeps alternating between more than two station lists would pile them up
        // (a few kB each, every iteration): beyond a handful the oldest go, after the device has finished everything enqueued --
        // a launch that still uses one of them was enqueued long before (the lists retired LAST stay)
        if (b->retired.size() >= 12) {
            HIPCHK(hipDeviceSynchronize());
            for (size_t i = 0; i + 6 < b->retired.size(); ++i) hipFree(b->retired[i]);
            b->retired.erase(b->retired.begin(), b->retired.end() - 6);
        }
        if (b->schur_map[slot]) b->retired.push_back(b->schur_map[slot]);
        if (b->schur_idx[slot]) b->retired.push_back(b->schur_idx[slot]);
        if (b->schur_spos[slot]) b->retired.push_back(b->schur_spos[slot]);
        b->schur_map[slot] = nullptr;
        b->schur_idx[slot] = nullptr;
        b->schur_spos[slot] = nullptr;
        HIPCHK(dnagpu::poison_malloc(&b->schur_map[slot], (size_t)npp * sizeof(int32_t)));
        HIPCHK(dnagpu::poison_malloc(&b->schur_idx[slot], k * sizeof(uint32_t)));
        HIPCHK(dnagpu::poison_malloc(&b->schur_spos[slot], (size_t)b->n_stn * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(b->schur_spos[slot], spos.data(), (size_t)b->n_stn * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->schur_map[slot], map.data(), (size_t)npp * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->schur_idx[slot], idx_out, k * sizeof(uint32_t), hipMemcpyHostToDevice));
        b->h_schur_idx[slot].assign(idx_out, idx_out + k);
        b->h_schur_nip[slot] = nip;
        b->h_schur_npp[slot] = npp;
    }
    *map_dev = b->schur_map[slot];
    *spos_dev = b->schur_spos[slot];
    *slot_out = slot;
    return DNAGPU_OK;
}

int schur_eliminate(dnagpu_ctx* ctx, int chain, Block* b, dnagpu_matrix* m, const uint32_t* idx_out, size_t k, const double** T, uint32_t* ldt,
                    int* slot_out, dnagpu_partial* keep = nullptr, const FormInOrder* form = nullptr) {
    const uint32_t n = form ? 3 * b->n_stn : m->n, nj = (uint32_t)(3 * k), ni = n - nj;
    uint32_t nip = ni ? pad128(ni) : 0, njp = pad128(nj + 1);
    // A light kept factor's capacity IS the shape the block is eliminated in (identity padding up to it): blocks of unequal size that
    // share a capacity can then go through the batched calls together, and a block gives the same bits batched or alone.
    if (keep && keep->spine && ni && keep->n_cap - keep->k_cap >= nip && keep->k_cap >= njp) {
        nip = keep->n_cap - keep->k_cap;
        njp = keep->k_cap;
    }
    const uint32_t npp = nip + njp;
    if (!form && (size_t)npp * nip > ((size_t)m->np_max + 128) * m->np_max) return fail(ctx, DNAGPU_EINVAL, "schur: matrix capacity");
    int slot = -1;
    const int32_t* map_dev = nullptr;
    const uint32_t* spos_dev = nullptr;
    {
        int rs = schur_order(ctx, b, idx_out, k, nip, nj, npp, &slot, &map_dev, &spos_dev);
        if (rs) return rs;
    }
    if (form && !keep) return fail(ctx, DNAGPU_EINVAL, "schur: forming in order needs a retained factor");
    int rc = ensure_ws(ctx, chain, npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    gemm_profile_close(ws);
    if (keep) {
        if (npp > keep->n_cap || njp > keep->k_cap) return fail(ctx, DNAGPU_EINVAL, "schur: retained factor capacity");
        hipStream_t st = ctx->stream[chain];
        keep->valid = false;
        keep->completed = false;
        keep->factored = false;
        keep->n = n; keep->nj = nj; keep->nip = nip; keep->njp = njp; keep->npp = npp;
        // the matrix being factored lives in the chain's X workspace (free here: the kept L^-1 has a home of its own); nothing of
        // it is needed after this call but the Schur complement, which the caller extracts at once
        double* F = ws.X;
        if (keep->store) keep->store->n = 0;       // (its storage now holds the factor's inverse, not a matrix)
        if (form) {
            uint32_t* dstn = nullptr;
            double* dw = nullptr;
            if (form->n_con) {
                for (size_t i = 0; i < form->n_con; ++i)
                    if (form->con_stn[i] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "schur: constraint station out of range");
                int rs = stage_u32(ctx, chain, form->con_stn, form->n_con, &dstn);
                if (!rs) rs = stage_f64(ctx, chain, form->con_w9, form->n_con * 9, &dw);
                if (rs) return rs;
            }
            HbmTimed timed(ctx, chain, DNAGPU_HBM_FORM_ORDERED, 4.0 * (double)npp * npp);
            launch_form_ordered(F, npp, npp, map_dev, spos_dev, b->pair_row, b->pair_col, b->pair_off, b->pair_ent, b->Wblk, b->n_pairs, b->n_wblk,
                                (uint32_t)chain * (b->n_tblk + b->n_dsblk), dstn, dw, (uint32_t)form->n_con, b->rhs[chain], nip + nj, st);
        } else {
            launch_schur_permute(m->F, m->np, map_dev, b->rhs[chain], F, npp, npp, st);
        }
        HIPCHK(hipMemcpyAsync(keep->map, map_dev, (size_t)npp * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        if (keep->spine)
            sym_spine_async(ws, F, keep->X, (int)npp, (int)(nip / 128), (int)(njp / 128));
        else
            sym_schur_keep_async(ws, F, keep->X, (int)npp, (int)(nip / 128), (int)(njp / 128));
        if (nip && !keep->spine)   // the panel L_KI (kept rows x eliminated columns) out of the chain's panel workspace
            HIPCHK(hipMemcpy2DAsync(keep->WK, (size_t)njp * sizeof(double), ws.W + nip, (size_t)npp * sizeof(double), (size_t)njp * sizeof(double), nip,
                                    hipMemcpyDeviceToDevice, st));
        // the row that carried the right-hand side rode along as a passenger: it is no unknown, its panel entries go
        if (nip && !keep->spine) HIPCHK(hipMemset2DAsync(keep->WK + nj, (size_t)njp * sizeof(double), 0, sizeof(double), nip, st));
        // (light form: the passenger row sits in the panels of X, row nip + nj of every block column; it is cleared there)
        if (nip && keep->spine) HIPCHK(hipMemset2DAsync(keep->X + nip + nj, (size_t)npp * sizeof(double), 0, sizeof(double), nip, st));
        keep->valid = true;
        *T = F + (size_t)nip * npp + nip;
    } else {
        launch_schur_permute(m->F, m->np, map_dev, b->rhs[chain], ws.W, npp, npp, ctx->stream[chain]);
        sym_schur_async(ws, ws.W, (int)npp, m->F, (int)npp, (int)(nip / 128), (int)(njp / 128));
        *T = ws.W + (size_t)nip * npp + nip;
    }
    *ldt = npp;
    *slot_out = slot;
    return DNAGPU_OK;
}
}  // namespace

namespace {
// forward half of the blocked substitution with a light factor (sym_inverse.h: sym_spine_async): for every diagonal block b of the
// eliminated part  v_b <- X_bb v_b,  v_below <- v_below - L_(below, b) v_b  (below: everything under the block, kept rows included)
void spine_forward(dnagpu_ctx* ctx, int chain, const dnagpu_partial* pf, double* rp) {
    hipStream_t st = ctx->stream[chain];
    const uint32_t ld = pf->npp;
    double* part = ctx->symv_part[chain];
    for (const auto& bl : sym_spine_blocks((int)(pf->nip / 128))) {
        const uint32_t o = (uint32_t)bl.first * 128, h = (uint32_t)bl.second * 128, below = ld - (o + h);
        launch_gemv(pf->X + (size_t)o * ld + o, ld, h, h, rp + o, part, SYMV_CHUNKS - 1, 1, nullptr, 1.0, rp + o, h, st);
        launch_gemv(pf->X + (size_t)o * ld + o + h, ld, below, h, rp + o, part, SYMV_CHUNKS - 1, 0, rp + o + h, -1.0, rp + o + h, below, st);
    }
}
}  // namespace

extern "C" {

int dnagpu_schur_carry(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m, const uint32_t* idx_out, size_t k, dnagpu_matrix* jm) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !m || !jm || !k || !idx_out || k > b->n_stn || 3 * k > jm->n_max || m->n != 3 * b->n_stn)
        return fail(ctx, DNAGPU_EINVAL, "schur_carry: bad arguments");
    const uint32_t nj = (uint32_t)(3 * k), npj = pad128(nj);
    int rc = ensure_symv(ctx, chain, npj);
    if (rc) return rc;
    const double* T = nullptr;
    uint32_t ldt = 0;
    int slot = 0;
    rc = schur_eliminate(ctx, chain, b, m, idx_out, k, &T, &ldt, &slot);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    jm->n = nj;
    jm->np = npj;
    if (ctx->info_carry) {
        // information form: the complement S (the junction stations' weight matrix) and the reduced right-hand side r travel as they are,
        // with the estimates they were formed at; the next block adds S to its normals and r + S (those estimates - its own) to its
        // right-hand side (dnagpu_junction_rhs) -- what the estimates x + S^-1 r weighted by S contribute, without S^-1: no inverse of
        // the complement (nj^3 flops and, for a 450-unknown junction, half of the step's launches)
        if (!jm->jrhs && dnagpu::poison_malloc(&jm->jrhs, (size_t)jm->np_max * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            jm->jrhs = nullptr;
            return fail(ctx, DNAGPU_ENOMEM, "schur_carry: right-hand side of the information form");
        }
        launch_schur_extract(T, ldt, nj, npj, jm->F, nullptr, jm->jrhs, st);
        launch_gather_vec3(b->x_est[chain], b->schur_idx[slot], (uint32_t)k, jm->jest, st);
        jm->form = 1;
        // (the elimination's verdict; a complement that is not positive definite shows in the block that receives it)
        HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    } else {
        // the complement IS the weight matrix of the junction stations; its inverse (their variances) gives their corrections
        launch_schur_extract(T, ldt, nj, npj, jm->F, m->F, ws.svec, st);
        sym_inverse_async(ws, m->F, nj, npj, false, /*reset_info=*/false);
        launch_symv(m->F, ws.svec, b->corr[chain], ctx->symv_part[chain], nj, npj, SYMV_CHUNKS, st);
        launch_schur_estimates(b->x_est[chain], b->schur_idx[slot], (uint32_t)k, b->corr[chain], jm->jest, st);
        jm->form = 0;
    }
    if (ws.hold_info) return DNAGPU_OK;          // (dnagpu_chain_hold_info: the verdict is taken once, after the run of steps)
    HIPCHK(hipStreamSynchronize(st));
    return check_info(ctx, chain);
}

int dnagpu_schur_carry_keep(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m, const uint32_t* idx_out, size_t k, dnagpu_matrix* jm,
                            dnagpu_partial* keep) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !m || !jm || !k || !idx_out || k >= b->n_stn || 3 * k > jm->n_max || m->n != 3 * b->n_stn || !keep || !keep->spine || keep->store)
        return fail(ctx, DNAGPU_EINVAL, "schur_carry_keep: bad arguments");
    if (!ctx->info_carry) return fail(ctx, DNAGPU_EINVAL, "schur_carry_keep: the information form of the carry is switched off");
    const uint32_t nj = (uint32_t)(3 * k), npj = pad128(nj);
    const double* T = nullptr;
    uint32_t ldt = 0;
    int slot = 0;
    int rc = schur_eliminate(ctx, chain, b, m, idx_out, k, &T, &ldt, &slot, keep);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    jm->n = nj;
    jm->np = npj;
    if (!jm->jrhs && dnagpu::poison_malloc(&jm->jrhs, (size_t)jm->np_max * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        jm->jrhs = nullptr;
        return fail(ctx, DNAGPU_ENOMEM, "schur_carry: right-hand side of the information form");
    }
    launch_schur_extract(T, ldt, nj, npj, jm->F, nullptr, jm->jrhs, st);
    launch_gather_vec3(b->x_est[chain], b->schur_idx[slot], (uint32_t)k, jm->jest, st);
    jm->form = 1;
    if (ws.hold_info) return DNAGPU_OK;          // (dnagpu_chain_hold_info; a failed run's factors are dropped by the caller)
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    rc = check_info(ctx, chain);
    if (rc) keep->valid = false;
    return rc;
}

int dnagpu_schur_carry_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, const uint32_t* idx_out, size_t k, dnagpu_matrix* jm, const dnagpu_partial* keep) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !jm || !k || !idx_out || !keep || !keep->spine || !keep->valid || keep->n != 3 * b->n_stn || keep->nj != 3 * k || jm->n != 3 * k ||
        jm->form != 1 || !jm->jrhs)
        return fail(ctx, DNAGPU_EINVAL, "schur_carry_rhs: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (idx_out[i] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "schur_carry_rhs: station out of range");
    int rc = ensure_ws(ctx, chain, keep->npp);
    if (!rc) rc = ensure_symv(ctx, chain, keep->npp);
    if (rc) return rc;
    uint32_t* didx = nullptr;
    rc = stage_u32(ctx, chain, idx_out, k, &didx);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    double* rp = ws.svec;
    launch_gather_map(b->rhs[chain], keep->map, keep->npp, rp, st);
    spine_forward(ctx, chain, keep, rp);
    HIPCHK(hipMemcpyAsync(jm->jrhs, rp + keep->nip, (size_t)keep->nj * sizeof(double), hipMemcpyDeviceToDevice, st));
    launch_gather_vec3(b->x_est[chain], didx, (uint32_t)k, jm->jest, st);
    // (stream ordered, no wait: the next step of this chain follows on the same stream, other chains read jm after the chains' phase has
    //  been synchronised)
    return DNAGPU_OK;
}

int dnagpu_chain_step_rhs(dnagpu_ctx* ctx, int chain, uint32_t rblk, uint32_t src_blk, const uint32_t* idx_keep, size_t k_keep, const dnagpu_matrix* red,
                          const dnagpu_matrix* jm_in, const uint32_t* idx_in, size_t k_in, dnagpu_matrix* jm_out, const uint32_t* idx_out, size_t k_out,
                          const dnagpu_partial* keep) {
    CHK_CTX();
    CHK_CHAIN();
    Block* rb = find_block(ctx, rblk);
    Block* sb = find_block(ctx, src_blk);
    if (!rb || !sb || !red || !idx_keep || rb->n_stn != k_keep || red->n != 3 * k_keep || !jm_out || !idx_out || !k_out || !keep || !keep->spine ||
        !keep->valid || keep->n != 3 * rb->n_stn || keep->nj != 3 * k_out || jm_out->n != 3 * k_out || jm_out->form != 1 || !jm_out->jrhs ||
        (jm_in && (!idx_in || jm_in->n != 3 * k_in || jm_in->form != 1 || !jm_in->jrhs)))
        return fail(ctx, DNAGPU_EINVAL, "chain_step_rhs: bad arguments");
    const std::vector<std::pair<int, int>> blocks = sym_spine_blocks((int)(keep->nip / 128));
    if (keep->npp > SMALL_STEP_MAX || 3 * k_keep > SMALL_STEP_MAX || (jm_in && 3 * k_in > SMALL_STEP_MAX) || blocks.size() > (size_t)SMALL_STEP_BLOCKS)
        return DNAGPU_ETOOLARGE;        // (not an error: the caller takes the step through the separate calls)
    // One workgroup streams the factor at one CU's share of the fabric (48 us per padded megabyte, profiles/HISTORY.md); the separate
    // calls are a dozen launches (~ 100 us) whose products run at the chip's HBM rate: beyond ~ 2 MB of factor they are the faster way
    // (cfg3's 2 048-unknown steps: 581 us in one launch).
    if (((size_t)keep->nip * keep->nip / 2 + (size_t)(keep->npp - keep->nip) * keep->nip) * sizeof(double) > SMALL_STEP_BYTES) return DNAGPU_ETOOLARGE;
    for (size_t i = 0; i < k_keep; ++i)
        if (idx_keep[i] >= sb->n_stn) return fail(ctx, DNAGPU_EINVAL, "chain_step_rhs: station out of range");
    for (size_t i = 0; i < k_out; ++i)
        if (idx_out[i] >= rb->n_stn) return fail(ctx, DNAGPU_EINVAL, "chain_step_rhs: station out of range");
    for (size_t i = 0; jm_in && i < k_in; ++i)
        if (idx_in[i] >= rb->n_stn) return fail(ctx, DNAGPU_EINVAL, "chain_step_rhs: station out of range");
    ChainRhsStep a{};
    uint32_t *dkeep = nullptr, *din = nullptr, *dout = nullptr;
    int rc = stage_u32(ctx, chain, idx_keep, k_keep, &dkeep);
    if (!rc) rc = stage_u32(ctx, chain, idx_out, k_out, &dout);
    if (!rc && jm_in) rc = stage_u32(ctx, chain, idx_in, k_in, &din);
    if (rc) return rc;
    // (three lists through ONE staging buffer would overwrite each other: all of them must have come from the cache)
    if (dkeep == ctx->scr_u32[chain] || dout == ctx->scr_u32[chain] || (jm_in && din == ctx->scr_u32[chain])) return DNAGPU_ETOOLARGE;
    gemm_profile_close(ctx->ws[chain]);
    a.red_rhs = red->jest; a.x_orig_src = sb->x_orig; a.keep_idx = dkeep; a.n_stn = rb->n_stn;
    a.rhs = rb->rhs[chain]; a.x_est = rb->x_est[chain];
    if (jm_in) {
        a.J = jm_in->F; a.npj = jm_in->np; a.jest_in = jm_in->jest; a.jrhs_in = jm_in->jrhs; a.idx_in = din; a.k_in = (uint32_t)k_in;
    }
    a.X = keep->X; a.map = keep->map; a.npp = keep->npp; a.nip = keep->nip; a.nj = keep->nj;
    a.nblocks = (int)blocks.size();
    for (size_t q = 0; q < blocks.size(); ++q) {
        a.blk_o[q] = (uint32_t)blocks[q].first * 128;
        a.blk_h[q] = (uint32_t)blocks[q].second * 128;
    }
    a.jrhs_out = jm_out->jrhs; a.jest_out = jm_out->jest; a.idx_out = dout; a.k_out = (uint32_t)k_out;
    launch_chain_rhs_step(a, ctx->stream[chain]);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

void dnagpu_small_batch_destroy(dnagpu_ctx* ctx, dnagpu_small_batch* sb) {
    if (!sb) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
    }
    if (sb->table) hipFree(sb->table);
    if (sb->result) hipFree(sb->result);
    if (sb->result_host) hipHostFree(sb->result_host);
    for (uint32_t* p : sb->idx_dev) hipFree(p);
    delete sb;
}

int dnagpu_small_batch_create(dnagpu_ctx* ctx, uint32_t n, const uint32_t* blks, dnagpu_partial* const* pf, dnagpu_matrix* const* red,
                              const dnagpu_matrix* const* j0, const uint32_t* const* idx0, const size_t* k0, const dnagpu_matrix* const* j1,
                              const uint32_t* const* idx1, const size_t* k1, const int* last, dnagpu_small_batch** out) {
    CHK_CTX();
    if (!out || !n || !blks || !pf || !red || !j0 || !idx0 || !k0 || !j1 || !idx1 || !k1 || !last)
        return fail(ctx, DNAGPU_EINVAL, "small_batch_create: bad arguments");
    *out = nullptr;
    std::vector<SmallBlockDesc> host(n);
    dnagpu_small_batch* sb = new (std::nothrow) dnagpu_small_batch();
    if (!sb) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    auto bail = [&](int rc) {
        dnagpu_small_batch_destroy(ctx, sb);
        return rc;
    };
    if (dnagpu::poison_malloc(&sb->result, (size_t)2 * n * sizeof(double)) != hipSuccess || hipHostMalloc(&sb->result_host, (size_t)2 * n * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        return bail(fail(ctx, DNAGPU_ENOMEM, "small_batch_create: result buffers"));
    }
    // the junction station lists of all blocks in one device buffer (one allocation, one copy: 1 300 of each were 35 ms for 666 blocks)
    std::vector<uint32_t> idx_all;
    std::vector<size_t> idx_at(2 * (size_t)n, 0);
    for (uint32_t q = 0; q < n; ++q) {
        const dnagpu_matrix* jm[2] = {j0[q], j1[q]};
        const uint32_t* ix[2] = {idx0[q], idx1[q]};
        const size_t kk[2] = {k0[q], k1[q]};
        for (int e = 0; e < 2; ++e) {
            idx_at[2 * (size_t)q + e] = idx_all.size();
            if (jm[e] && ix[e]) idx_all.insert(idx_all.end(), ix[e], ix[e] + kk[e]);
        }
    }
    uint32_t* idx_dev_all = nullptr;
    if (!idx_all.empty()) {
        if (dnagpu::poison_malloc(&idx_dev_all, idx_all.size() * sizeof(uint32_t)) != hipSuccess ||
            hipMemcpy(idx_dev_all, idx_all.data(), idx_all.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (idx_dev_all) hipFree(idx_dev_all);
            return bail(fail(ctx, DNAGPU_ENOMEM, "small_batch_create: station lists"));
        }
        sb->idx_dev.push_back(idx_dev_all);
    }
    for (uint32_t q = 0; q < n; ++q) {
        Block* b = find_block(ctx, blks[q]);
        const dnagpu_partial* p = pf[q];
        if (!b || !p || !red[q]) return bail(fail(ctx, DNAGPU_EINVAL, "small_batch_create: bad arguments"));
        // (limits of the one-workgroup kernels; terrestrial rows and direction sets have no place in a reuse iteration)
        const std::vector<std::pair<int, int>> blocks = sym_spine_blocks((int)(p->nip / 128));
        if (!p->spine || !p->factored || p->n != 3 * b->n_stn || red[q]->n != p->nj || p->npp > SMALL_STEP_MAX || 3 * b->n_stn > SMALL_STEP_MAX ||
            blocks.size() > (size_t)SMALL_STEP_BLOCKS || b->n_t || b->n_dsblk)
            return bail(DNAGPU_ETOOLARGE);
        if (last[q] && !b->corr_keep && dnagpu::poison_malloc(&b->corr_keep, (size_t)b->n_stn * 3 * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            b->corr_keep = nullptr;
            return bail(fail(ctx, DNAGPU_ENOMEM, "small_batch_create: corrections set aside"));
        }
        SmallBlockDesc& d = host[q];
        memset(&d, 0, sizeof(d));
        d.wblk = b->Wblk; d.vec_wrow = b->vec_wrow; d.vec_c0 = b->vec_c0; d.vec_k = b->vec_k; d.n_vec = b->n_bl;
        d.inc_off = b->inc_off; d.inc = b->inc; d.b = b->b[0];
        d.wb = b->wb[0]; d.rhs = b->rhs[0]; d.corr = b->corr[0]; d.corr_keep = last[q] ? b->corr_keep : nullptr;
        d.x_orig = b->x_orig; d.x_est = b->x_est[0]; d.x_rig = b->x_rig; d.n_stn = b->n_stn;
        d.X = p->X; d.map = p->map; d.npp = p->npp; d.nip = p->nip; d.nj = p->nj;
        d.nblocks = (int)blocks.size();
        for (size_t e = 0; e < blocks.size(); ++e) {
            d.blk_o[e] = (uint32_t)blocks[e].first * 128;
            d.blk_h[e] = (uint32_t)blocks[e].second * 128;
        }
        d.red_rhs = red[q]->jest;
        const dnagpu_matrix* jm[2] = {j0[q], j1[q]};
        const uint32_t* ix[2] = {idx0[q], idx1[q]};
        const size_t kk[2] = {k0[q], k1[q]};
        for (int e = 0; e < 2; ++e) {
            if (!jm[e]) continue;
            if (!ix[e] || jm[e]->n != 3 * kk[e] || jm[e]->form != 1 || !jm[e]->jrhs || 3 * kk[e] > SMALL_STEP_MAX) return bail(DNAGPU_ETOOLARGE);
            for (size_t i = 0; i < kk[e]; ++i)
                if (ix[e][i] >= b->n_stn) return bail(fail(ctx, DNAGPU_EINVAL, "small_batch_create: station out of range"));
            uint32_t* dev = idx_dev_all + idx_at[2 * (size_t)q + e];
            d.J[e] = jm[e]->F; d.jest[e] = jm[e]->jest; d.jrhs[e] = jm[e]->jrhs; d.jidx[e] = dev; d.jk[e] = (uint32_t)kk[e]; d.jnp[e] = jm[e]->np;
        }
        d.last = last[q] ? 1u : 0u;
        d.result = sb->result + 2 * (size_t)q;
    }
    if (dnagpu::poison_malloc(&sb->table, (size_t)n * sizeof(SmallBlockDesc)) != hipSuccess ||
        hipMemcpy(sb->table, host.data(), (size_t)n * sizeof(SmallBlockDesc), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return bail(fail(ctx, DNAGPU_ENOMEM, "small_batch_create: block table"));
    }
    sb->n = n;
    *out = sb;
    return DNAGPU_OK;
}

int dnagpu_small_batch_condense(dnagpu_ctx* ctx, int chain, dnagpu_small_batch* sb) {
    CHK_CTX();
    CHK_CHAIN();
    if (!sb || !sb->n) return fail(ctx, DNAGPU_EINVAL, "small_batch_condense: bad arguments");
    gemm_profile_close(ctx->ws[chain]);
    launch_small_condense((const SmallBlockDesc*)sb->table, sb->n, ctx->stream[chain]);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

int dnagpu_small_batch_solve(dnagpu_ctx* ctx, int chain, dnagpu_small_batch* sb, double* max_corr) {
    CHK_CTX();
    CHK_CHAIN();
    if (!sb || !sb->n || !max_corr) return fail(ctx, DNAGPU_EINVAL, "small_batch_solve: bad arguments");
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ctx->ws[chain]);
    launch_small_solve((const SmallBlockDesc*)sb->table, sb->n, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sb->result_host, sb->result, (size_t)2 * sb->n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint32_t q = 0; q < sb->n; ++q) max_corr[q] = sb->result_host[2 * (size_t)q];
    return DNAGPU_OK;
}

int dnagpu_partial_create(dnagpu_ctx* ctx, uint32_t n_max, uint32_t k_max, dnagpu_partial** out) {
    CHK_CTX();
    if (!out || !k_max || k_max > n_max) return fail(ctx, DNAGPU_EINVAL, "partial_create: bad arguments");
    *out = nullptr;
    dnagpu_partial* p = new (std::nothrow) dnagpu_partial();
    if (!p) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    p->k_cap = pad128(k_max + 1);
    p->n_cap = pad128(n_max - k_max ? n_max - k_max : 1) + p->k_cap;
    hipError_t e = dnagpu::poison_malloc(&p->X, (size_t)p->n_cap * p->n_cap * sizeof(double));
    if (e == hipSuccess) e = dnagpu::poison_malloc(&p->WK, (size_t)p->k_cap * p->n_cap * sizeof(double));
    if (e == hipSuccess) e = dnagpu::poison_malloc(&p->map, (size_t)p->n_cap * sizeof(int32_t));
    if (e != hipSuccess) {
        dnagpu_partial_destroy(ctx, p);
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "retained factor allocation", e);
    }
    *out = p;
    return DNAGPU_OK;
}

int dnagpu_partial_create_in(dnagpu_ctx* ctx, uint32_t n_max, uint32_t k_max, dnagpu_matrix* store, dnagpu_partial** out) {
    CHK_CTX();
    if (!out || !store || !k_max || k_max > n_max) return fail(ctx, DNAGPU_EINVAL, "partial_create_in: bad arguments");
    *out = nullptr;
    dnagpu_partial* p = new (std::nothrow) dnagpu_partial();
    if (!p) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    p->k_cap = pad128(k_max + 1);
    p->n_cap = pad128(n_max - k_max ? n_max - k_max : 1) + p->k_cap;
    if ((size_t)p->n_cap * p->n_cap > ((size_t)store->np_max + 128) * store->np_max) {
        delete p;
        return fail(ctx, DNAGPU_EINVAL, "partial_create_in: the matrix is too small (create it with n_max + 256)");
    }
    p->store = store;
    p->X = store->F;
    hipError_t e = dnagpu::poison_malloc(&p->WK, (size_t)p->k_cap * p->n_cap * sizeof(double));
    if (e == hipSuccess) e = dnagpu::poison_malloc(&p->map, (size_t)p->n_cap * sizeof(int32_t));
    if (e != hipSuccess) {
        dnagpu_partial_destroy(ctx, p);
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "retained factor allocation", e);
    }
    *out = p;
    return DNAGPU_OK;
}

int dnagpu_partial_create_spine(dnagpu_ctx* ctx, uint32_t n_max, uint32_t k_max, dnagpu_matrix* store, dnagpu_partial** out) {
    CHK_CTX();
    if (!out || !k_max || k_max > n_max) return fail(ctx, DNAGPU_EINVAL, "partial_create_spine: bad arguments");
    *out = nullptr;
    dnagpu_partial* p = new (std::nothrow) dnagpu_partial();
    if (!p) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    p->k_cap = pad128(k_max + 1);
    p->n_cap = pad128(n_max - k_max ? n_max - k_max : 1) + p->k_cap;
    if (store && (size_t)p->n_cap * p->n_cap > ((size_t)store->np_max + 128) * store->np_max) {
        delete p;
        return fail(ctx, DNAGPU_EINVAL, "partial_create_spine: the matrix is too small (create it with n_max + 256)");
    }
    p->store = store;
    p->spine = true;
    hipError_t e = hipSuccess;
    if (store)
        p->X = store->F;
    else
        e = dnagpu::poison_malloc(&p->X, (size_t)p->n_cap * p->n_cap * sizeof(double));
    if (e == hipSuccess) e = dnagpu::poison_malloc(&p->map, (size_t)p->n_cap * sizeof(int32_t));
    if (e != hipSuccess) {
        dnagpu_partial_destroy(ctx, p);
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "retained factor allocation", e);
    }
    *out = p;
    return DNAGPU_OK;
}

void dnagpu_partial_destroy(dnagpu_ctx* ctx, dnagpu_partial* p) {
    if (!p) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
    }
    if (p->X && !p->store) hipFree(p->X);
    if (p->WK) hipFree(p->WK);
    if (p->map) hipFree(p->map);
    delete p;
}

int dnagpu_partial_complete(dnagpu_ctx* ctx, int chain, dnagpu_partial* pf, const dnagpu_matrix* kk, dnagpu_matrix* inv) {
    CHK_CTX();
    CHK_CHAIN();
    if (!pf || !pf->valid || !kk || !inv || kk->n != pf->nj || pf->n > inv->n_max)
        return fail(ctx, DNAGPU_EINVAL, "partial_complete: bad arguments");
    if (pf->spine) {
        int rc2 = dnagpu_partial_complete_factor(ctx, chain, pf, kk);
        return rc2 ? rc2 : dnagpu_partial_finish(ctx, chain, pf, inv);
    }
    int rc = ensure_ws(ctx, chain, pf->npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    pf->valid = false;       // consumed
    pf->completed = pf->store == nullptr;      // (borrowed storage goes back to its matrix: nothing left for dnagpu_partial_reduce_rhs)
    // the matrix being completed lives in the chain's X workspace: only its trailing block (set here) is read, and the result
    // leaves it through the un-permutation -- which may overwrite the kept L^-1 (inv == the matrix that lent its storage)
    double* F = ws.X;
    launch_partial_set_trailing(F + (size_t)pf->nip * pf->npp + pf->nip, pf->npp, pf->njp, kk->F, kk->np, pf->nj, st);
    sym_complete_async(ws, F, pf->X, (int)pf->npp, pf->WK, (int)pf->njp, (int)(pf->nip / 128), (int)(pf->njp / 128));
    inv->n = pf->n;
    inv->np = pad128(pf->n);
    launch_init_padding(inv->F, inv->n, inv->np, st);      // (the un-permutation writes every element of the n x n part)
    {
        HbmTimed t(ctx, chain, DNAGPU_HBM_UNPERMUTE, 16.0 * (double)inv->np * inv->np);
        launch_unpermute(F, pf->npp, pf->npp, pf->map, inv->F, inv->np, st);
    }
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info(ctx, chain);
}

int dnagpu_partial_complete_factor(dnagpu_ctx* ctx, int chain, dnagpu_partial* pf, const dnagpu_matrix* kk) {
    CHK_CTX();
    CHK_CHAIN();
    if (!pf || !pf->valid || !kk || kk->n != pf->nj) return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor: bad arguments");
    int rc = ensure_ws(ctx, chain, pf->npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    pf->valid = false;
    pf->completed = false;
    double* F = ws.X;        // scratch: the kept block's factorisation and the T_KI panel pass through it
    launch_partial_set_trailing(F + (size_t)pf->nip * pf->npp + pf->nip, pf->npp, pf->njp, kk->F, kk->np, pf->nj, st);
    if (pf->spine)
        sym_spine_kept_async(ws, F, pf->X, (int)pf->npp, (int)(pf->nip / 128), (int)(pf->njp / 128));
    else
        sym_complete_async(ws, F, pf->X, (int)pf->npp, pf->WK, (int)pf->njp, (int)(pf->nip / 128), (int)(pf->njp / 128), 1);
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    rc = check_info(ctx, chain);
    pf->factored = rc == DNAGPU_OK;
    return rc;
}

int dnagpu_partial_solve(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_partial* pf) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !pf || !pf->factored || 3 * b->n_stn != pf->n) return fail(ctx, DNAGPU_EINVAL, "partial_solve: bad arguments");
    int rc = ensure_ws(ctx, chain, pf->npp);
    if (!rc) rc = ensure_symv(ctx, chain, pf->npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    // corrections = N^-1 rhs = P (X^T (X (P^T rhs))): two triangular matrix-vector products in the elimination's order
    double* rp = ws.svec;                                           // P^T rhs, then the result
    double* y = ctx->symv_part[chain] + (size_t)SYMV_CHUNKS * pf->npp - pf->npp;     // (the last chunk row of the partial sums: free once they are added up)
    // (HBM side: the factor -- lower block triangle, npp^2 / 2 doubles -- is read twice)
    HbmTimed timed(ctx, chain, DNAGPU_HBM_SUBSTITUTION, 8.0 * (double)pf->npp * pf->npp);
    launch_gather_map(b->rhs[chain], pf->map, pf->npp, rp, st);
    if (pf->spine) {
        // blocked substitution with the block factor: forward  y_b = X_bb v_b,  v_below -= L_(below, b) y_b;  the kept block
        // v_K = X_KK^T (X_KK v_K);  backward  v_b = X_bb^T (y_b - L_(below, b)^T v_below).  ~n^2 doubles read twice, a few launches per block.
        const uint32_t ld = pf->npp;
        const std::vector<std::pair<int, int>> blocks = sym_spine_blocks((int)(pf->nip / 128));
        double* part = ctx->symv_part[chain];
        spine_forward(ctx, chain, pf, rp);
        {
            const uint32_t o = pf->nip, h = pf->njp;
            launch_gemv(pf->X + (size_t)o * ld + o, ld, h, h, rp + o, part, SYMV_CHUNKS - 1, 1, nullptr, 1.0, rp + o, h, st);
            launch_gemv_t_lower(pf->X + (size_t)o * ld + o, ld, h, rp + o, y, st);
            HIPCHK(hipMemcpyAsync(rp + o, y, (size_t)h * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        for (size_t q = blocks.size(); q-- > 0;) {
            const uint32_t o = (uint32_t)blocks[q].first * 128, h = (uint32_t)blocks[q].second * 128, below = ld - (o + h);
            launch_gemv_t(pf->X + (size_t)o * ld + o + h, ld, below, h, rp + o + h, rp + o, -1.0, y, st);
            launch_gemv_t_lower(pf->X + (size_t)o * ld + o, ld, h, y, rp + o, st);
        }
        launch_scatter_map(rp, pf->map, pf->npp, b->corr[chain], st);
        return DNAGPU_OK;
    }
    launch_gemv(pf->X, pf->npp, pf->npp, pf->npp, rp, ctx->symv_part[chain], SYMV_CHUNKS - 1, 1, nullptr, 1.0, y, pf->npp, st);
    launch_gemv_t_lower(pf->X, pf->npp, pf->npp, y, rp, st);
    launch_scatter_map(rp, pf->map, pf->npp, b->corr[chain], st);
    return DNAGPU_OK;
}

int dnagpu_partial_finish(dnagpu_ctx* ctx, int chain, dnagpu_partial* pf, dnagpu_matrix* inv) {
    CHK_CTX();
    CHK_CHAIN();
    if (!pf || !pf->factored || !inv || pf->n > inv->n_max) return fail(ctx, DNAGPU_EINVAL, "partial_finish: bad arguments");
    int rc = ensure_ws(ctx, chain, pf->npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    pf->factored = false;
    pf->completed = pf->store == nullptr;
    double* F = ws.X;
    if (pf->spine) {
        pf->completed = false;
        sym_spine_finish_async(ws, F, pf->X, (int)pf->npp, (int)(pf->nip / 128), (int)(pf->njp / 128));
    } else {
        sym_complete_async(ws, F, pf->X, (int)pf->npp, pf->WK, (int)pf->njp, (int)(pf->nip / 128), (int)(pf->njp / 128), 2);
    }
    inv->n = pf->n;
    inv->np = pad128(pf->n);
    launch_init_padding(inv->F, inv->n, inv->np, st);      // (the un-permutation writes every element of the n x n part)
    {
        HbmTimed t(ctx, chain, DNAGPU_HBM_UNPERMUTE, 16.0 * (double)inv->np * inv->np);
        launch_unpermute(F, pf->npp, pf->npp, pf->map, inv->F, inv->np, st);
    }
    // (nothing here factors anything: `info` is put to "no failure" for check_info, which then reports enqueue / launch errors only)
    HIPCHK(hipMemsetAsync(ws.info, 0x7f, sizeof(int), st));
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info(ctx, chain);
}

int dnagpu_partial_reduce_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_partial* pf, dnagpu_matrix* red) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !pf || !red || red->n != pf->nj || 3 * b->n_stn != pf->n || !(pf->spine ? (pf->valid || pf->factored) : pf->completed))
        return fail(ctx, DNAGPU_EINVAL, "partial_reduce_rhs: bad arguments");
    int rc = ensure_ws(ctx, chain, pf->npp);
    if (!rc) rc = ensure_symv(ctx, chain, pf->npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    double* rp = ws.svec;                 // rhs in the elimination's order
    if (pf->spine) {
        // light form: the forward half of dnagpu_partial_solve's blocked substitution -- y_b = X_bb v_b, v_below -= L_(below, b) y_b over
        // the eliminated part's diagonal blocks; the kept rows ride along in every panel and end as the reduced right-hand side
        HbmTimed timed(ctx, chain, DNAGPU_HBM_SUBSTITUTION, 4.0 * (double)pf->npp * pf->npp);
        launch_gather_map(b->rhs[chain], pf->map, pf->npp, rp, st);
        spine_forward(ctx, chain, pf, rp);
        HIPCHK(hipMemcpyAsync(red->jest, rp + pf->nip, (size_t)pf->nj * sizeof(double), hipMemcpyDeviceToDevice, st));
        // (stream ordered, no wait: nothing here can fail on the device, and whoever reads red on another chain does so after the
        //  phase's chains have been synchronised -- dna_adjust::OnEveryChain)
        return DNAGPU_OK;
    }
    double* y = b->corr[chain];           // L_II^-1 rhs_I (n_i <= 3 n_stn values)
    launch_gather_map(b->rhs[chain], pf->map, pf->npp, rp, st);
    const uint32_t ni = pf->n - pf->nj;
    launch_gemv(pf->X, pf->npp, ni, ni, rp, ctx->symv_part[chain], SYMV_CHUNKS, 1, nullptr, 1.0, y, ni, st);
    launch_gemv(pf->WK, pf->njp, pf->njp, ni, y, ctx->symv_part[chain], SYMV_CHUNKS, 0, rp + pf->nip, -1.0, red->jest, pf->nj, st);
    HIPCHK(hipStreamSynchronize(st));     // red is read next by whichever chain runs the condensed block
    return DNAGPU_OK;
}

int dnagpu_block_reduce(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m, const uint32_t* idx_keep, size_t k, dnagpu_matrix* red,
                        dnagpu_partial* keep) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !m || !red || !k || !idx_keep || k > b->n_stn || 3 * k > red->n_max || m->n != 3 * b->n_stn)
        return fail(ctx, DNAGPU_EINVAL, "block_reduce: bad arguments");
    const double* T = nullptr;
    uint32_t ldt = 0;
    int slot = 0;
    int rc = schur_eliminate(ctx, chain, b, m, idx_keep, k, &T, &ldt, &slot, keep);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    red->n = (uint32_t)(3 * k);
    red->np = pad128(red->n);
    launch_schur_extract(T, ldt, red->n, red->np, red->F, m->F, red->jest, st);
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info(ctx, chain);
}

int dnagpu_block_form_reduce(dnagpu_ctx* ctx, int chain, uint32_t blk, const uint32_t* con_stn, const double* con_w9, size_t n_con,
                             const uint32_t* idx_keep, size_t k, dnagpu_matrix* red, dnagpu_partial* keep) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !red || !keep || !k || !idx_keep || k > b->n_stn || 3 * k > red->n_max || (n_con && (!con_stn || !con_w9)))
        return fail(ctx, DNAGPU_EINVAL, "block_form_reduce: bad arguments");
    const double* T = nullptr;
    uint32_t ldt = 0;
    int slot = 0;
    FormInOrder form{con_stn, con_w9, n_con};
    int rc = schur_eliminate(ctx, chain, b, nullptr, idx_keep, k, &T, &ldt, &slot, keep, &form);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    red->n = (uint32_t)(3 * k);
    red->np = pad128(red->n);
    launch_schur_extract(T, ldt, red->n, red->np, red->F, nullptr, red->jest, st);
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info(ctx, chain);
}

// ---- batched forms of the large steps of the condensed schedule (include/dnagpu.h) -------------------------------------------------
}  // extern "C"

namespace {

// widest diagonal block of the spine of ti eliminated tiles with tj kept ones: what a member's panel buffer must hold
uint32_t batch_panel_cols(uint32_t nip, uint32_t njp) {
    uint32_t h = njp / 128;
    for (const auto& bl : sym_spine_blocks((int)(nip / 128))) h = std::max<uint32_t>(h, (uint32_t)bl.second);
    return h * 128;
}

// workspaces of members 1 .. nb - 1 on this chain (member 0 works in the chain's own X / W)
int ensure_batch_ws(dnagpu_ctx* ctx, int chain, int nb, uint32_t npp, uint32_t wcols) {
    int rc = ensure_ws(ctx, chain, npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    if (nb > 1 && g_fail_batch_ws.load() > 0 && g_fail_batch_ws.fetch_sub(1) > 0) {
        HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
        for (int q = 1; q < BATCH_MAX; ++q) {
            if (ws.bX[q]) hipFree(ws.bX[q]);
            if (ws.bW[q]) hipFree(ws.bW[q]);
            ws.bX[q] = ws.bW[q] = nullptr;
        }
        return fail(ctx, DNAGPU_ENOMEM, "batch workspace allocation (injected)");
    }
    if (ws.bnp_cap < npp || ws.bw_cols < wcols) {
        HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
        for (int b = 0; b < BATCH_MAX; ++b) {
            if (ws.bX[b]) hipFree(ws.bX[b]);
            if (ws.bW[b]) hipFree(ws.bW[b]);
            ws.bX[b] = ws.bW[b] = nullptr;
        }
        ws.bnp_cap = std::max(ws.bnp_cap, npp);
        ws.bw_cols = std::max(ws.bw_cols, wcols);
    }
    for (int b = 1; b < nb; ++b) {
        hipError_t e = hipSuccess;
        if (!ws.bX[b]) e = dnagpu::poison_malloc(&ws.bX[b], (size_t)ws.bnp_cap * ws.bnp_cap * sizeof(double));
        if (e == hipSuccess && !ws.bW[b]) e = dnagpu::poison_malloc(&ws.bW[b], ((size_t)ws.bw_cols + 128) * ws.bnp_cap * sizeof(double));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // nothing half allocated stays behind: the caller falls back to one member at a time and needs the memory for that
            for (int q = 1; q < BATCH_MAX; ++q) {
                if (ws.bX[q]) hipFree(ws.bX[q]);
                if (ws.bW[q]) hipFree(ws.bW[q]);
                ws.bX[q] = ws.bW[q] = nullptr;
            }
            return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "batch workspace allocation", e);
        }
    }
    return DNAGPU_OK;
}

// the batch of a call on `chain`: F = the members' matrices being factored (the chain's X and bX), X = their kept factors, P = panels
void set_batch(InvWorkspace& ws, int nb, uint32_t npp, dnagpu_partial* const* pf) {
    InvBatch& bt = ws.batch;
    bt = InvBatch();
    bt.nb = nb;
    double* F[BATCH_MAX];
    double* X[BATCH_MAX];
    double* P[BATCH_MAX];
    for (int b = 0; b < nb; ++b) {
        F[b] = b ? ws.bX[b] : ws.X;
        X[b] = pf[b]->X;
        P[b] = b ? ws.bW[b] : ws.W;
    }
    bt.add(F[0], (size_t)npp * npp, F);
    bt.add(X[0], (size_t)npp * npp, X);
    bt.add(P[0], ((size_t)ws.bw_cols + 128) * npp, P);
}

// info of every member after a batched call (stream synchronised): the first failing member is reported like check_info does
int check_info_batch(dnagpu_ctx* ctx, int chain, int nb, int* failed_member) {
    InvWorkspace& ws = ctx->ws[chain];
    if (failed_member) *failed_member = -1;
    for (int b = 1; b < nb; ++b)
        if (ws.info_host[b] != INFO_SENTINEL && ws.info_host[0] == INFO_SENTINEL) {
            ws.info_host[0] = ws.info_host[b];
            if (failed_member) *failed_member = b;
            break;
        }
    int rc = check_info(ctx, chain);
    if (rc == DNAGPU_ENOTPOSDEF && failed_member && *failed_member < 0) *failed_member = 0;
    return rc;
}

}  // namespace

extern "C" {

// ---- chain plans (include/dnagpu.h; kernels: small_steps.hip) ---------------------------------------------------------------------------
void dnagpu_chain_plan_destroy(dnagpu_ctx* ctx, dnagpu_chain_plan* plan) {
    if (!plan) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
    }
    if (plan->table) hipFree(plan->table);
    if (plan->blob) hipFree(plan->blob);
    if (plan->factors) hipFree(plan->factors);
    delete plan;
}

int dnagpu_chain_plan_create(dnagpu_ctx* ctx, size_t n_steps, const dnagpu_chain_step* steps, size_t n_batches, const uint32_t* batch_first,
                             double max_bytes, dnagpu_chain_plan** out) {
    CHK_CTX();
    if (!out) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: null out");
    *out = nullptr;
    if (!n_steps || !steps || !n_batches || !batch_first || batch_first[0] != 0 || batch_first[n_batches] != n_steps)
        return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad arguments");
    // (whatever the plan has allocated on the device goes with it on every error path below: its factors alone can be 96 GB)
    struct PlanDeleter {
        void operator()(dnagpu_chain_plan* p) const {
            if (!p) return;
            for (void* q : {p->table, p->blob, (void*)p->factors})
                if (q) hipFree(q);
            delete p;
        }
    };
    std::unique_ptr<dnagpu_chain_plan, PlanDeleter> plan(new (std::nothrow) dnagpu_chain_plan());
    if (!plan) return fail(ctx, DNAGPU_ENOMEM, "host allocation");
    // a junction's right-hand side, allocated on first use: no memory for it is DNAGPU_ENOMEM (the caller then runs the chains step by step)
    auto ensure_jrhs = [&](dnagpu_matrix* m) -> int {
        if (m->jrhs) return DNAGPU_OK;
        hipError_t e = dnagpu::poison_malloc(&m->jrhs, (size_t)m->np_max * sizeof(double));
        if (e == hipSuccess) return DNAGPU_OK;
        (void)hipGetLastError();
        return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "chain_plan_create: junction right-hand side", e);
    };
    plan->n_steps = n_steps;
    plan->batch_first.assign(batch_first, batch_first + n_batches + 1);
    plan->shape.resize(n_batches);
    plan->factored.assign(n_batches, 0);
    plan->X.resize(n_steps);
    plan->out.resize(n_steps);
    std::vector<CbStep> table(n_steps);
    // everything the steps refer to that is not a matrix goes into one blob: offsets first, device addresses once it is allocated
    std::vector<uint8_t> blob;
    auto put = [&](const void* src, size_t bytes) {
        const size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + bytes);
        if (src) memcpy(blob.data() + off, src, bytes);
        return off;
    };
    struct Offs { size_t est, con, map, keep, xe, rhs, pos[CB_SRC_MAX], inv[CB_SRC_MAX]; bool has_est, has_con, has_map; };
    std::vector<Offs> offs(n_steps);
    double factor_bytes = 0.0;
    std::vector<size_t> x_off(n_steps);
    for (size_t q = 0; q < n_batches; ++q) {
        const uint32_t f = batch_first[q], l = batch_first[q + 1];
        if (l <= f || l - f > (uint32_t)BATCH_MAX) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: a batch is empty or too large");
        uint32_t nip = 0, njp = 0, outnp_max = 0;
        bool matrix_only = false;
        for (uint32_t s = f; s < l; ++s) {
            const dnagpu_chain_step& st = steps[s];
            if (s == f) matrix_only = st.matrix_only != 0;
            if ((st.matrix_only != 0) != matrix_only) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: a batch mixes steps of two kinds");
            if (matrix_only && l - f > (uint32_t)BLOCK_BATCH_MAX) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: a batch of kept blocks is too large");
            if (matrix_only) {
                if (!st.n_stn || st.n_src < 1 || st.n_src > CB_SRC_MAX || (st.n_con && (!st.con_stn || !st.con_w9))) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad step");
                if (3 * st.n_stn > SMALL_STEP_MAX) return DNAGPU_ETOOLARGE;
                continue;
            }
            const uint32_t n = 3 * st.n_stn, nj = (uint32_t)(3 * st.n_keep);
            if (!st.n_stn || !st.n_keep || !st.keep || !st.out || st.n_src < 1 || st.n_src > CB_SRC_MAX || (st.n_con && (!st.con_stn || !st.con_w9)) ||
                (st.est_blk == nullptr) != (st.est_idx == nullptr) || nj > st.out->n_max)
                return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad step");
            if (nj > n) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: more stations kept than there are");
            if (n > SMALL_STEP_MAX) return DNAGPU_ETOOLARGE;      // (nj = n: nothing to eliminate -- the system itself is carried on)
            nip = std::max(nip, pad128(n - nj));
            njp = std::max(njp, pad128(nj + 1));
            outnp_max = std::max(outnp_max, pad128(nj));
        }
        const uint32_t npp = nip + njp;
        if (matrix_only) {
            plan->shape[q] = {0, 0, 0, 0};
            for (uint32_t s = f; s < l; ++s) x_off[s] = 0;
            continue;
        }
        if (npp > SMALL_STEP_MAX || sym_spine_blocks((int)(nip / 128)).size() > (size_t)SMALL_STEP_BLOCKS) return DNAGPU_ETOOLARGE;
        plan->shape[q] = {nip, njp, npp, outnp_max};
        for (uint32_t s = f; s < l; ++s) {
            x_off[s] = (size_t)(factor_bytes / 8.0);
            factor_bytes += 8.0 * (double)npp * npp;
        }
    }
    // beyond the budget the plan keeps no factors: every run of a batch eliminates again, into scratch of the chain it runs on
    plan->keeps = factor_bytes <= max_bytes;
    plan->factor_bytes = plan->keeps ? factor_bytes : 0.0;
    for (size_t q = 0; q < n_batches; ++q) {
        const dnagpu_chain_plan::Shape sh = plan->shape[q];
        const auto blocks = sym_spine_blocks((int)(sh.nip / 128));
        for (uint32_t s = batch_first[q]; s < batch_first[q + 1]; ++s) {
            const dnagpu_chain_step& st = steps[s];
            CbStep& d = table[s];
            Offs& o = offs[s];
            memset(&d, 0, sizeof(d));
            const bool mo = st.matrix_only != 0;
            const uint32_t n = 3 * st.n_stn, nj = mo ? n : (uint32_t)(3 * st.n_keep);
            d.n_stn = st.n_stn; d.nj = nj; d.k_out = mo ? 0u : (uint32_t)st.n_keep; d.nip = sh.nip; d.njp = sh.njp; d.npp = sh.npp; d.n_src = (uint32_t)st.n_src;
            o.has_map = !mo;
            d.nblocks = (int)blocks.size();
            for (size_t b = 0; b < blocks.size(); ++b) {
                d.blk_o[b] = (uint32_t)blocks[b].first * 128;
                d.blk_h[b] = (uint32_t)blocks[b].second * 128;
            }
            // the elimination's order: the stations that leave (system order), padding, the kept stations (list order), the right-hand side's row
            std::vector<uint8_t> kept(st.n_stn, 0);
            for (size_t i = 0; i < st.n_keep && !mo; ++i) {
                if (st.keep[i] >= st.n_stn || kept[st.keep[i]]) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad list of kept stations");
                kept[st.keep[i]] = 1;
            }
            o.map = o.keep = o.xe = o.rhs = 0;
            if (!mo) {
                std::vector<int32_t> map(sh.npp, -1);
                uint32_t pos = 0;
                for (uint32_t t = 0; t < st.n_stn; ++t)
                    if (!kept[t])
                        for (int c = 0; c < 3; ++c) map[pos++] = (int32_t)(3 * t + c);
                for (size_t i = 0; i < st.n_keep; ++i)
                    for (int c = 0; c < 3; ++c) map[sh.nip + 3 * i + c] = (int32_t)(3 * st.keep[i] + c);
                map[sh.nip + nj] = -2;
                o.map = put(map.data(), map.size() * sizeof(int32_t));
                o.keep = put(st.keep, st.n_keep * sizeof(uint32_t));
                o.xe = put(nullptr, (size_t)n * sizeof(double));
                o.rhs = put(nullptr, (size_t)n * sizeof(double));
            }
            o.has_est = !mo && st.est_blk != nullptr;
            if (o.has_est) {
                std::vector<const double*> ptr(st.n_stn);
                for (uint32_t t = 0; t < st.n_stn; ++t) {
                    Block* b = find_block(ctx, st.est_blk[t]);
                    if (!b || st.est_idx[t] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad linearisation point");
                    ptr[t] = b->x_orig + (size_t)3 * st.est_idx[t];
                }
                o.est = put(ptr.data(), ptr.size() * sizeof(const double*));
            }
            o.has_con = st.n_con != 0;
            if (o.has_con) {
                std::vector<double> con((size_t)9 * st.n_stn, 0.0);
                for (size_t i = 0; i < st.n_con; ++i) {
                    if (st.con_stn[i] >= st.n_stn) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: constraint station out of range");
                    for (int e = 0; e < 9; ++e) con[(size_t)9 * st.con_stn[i] + e] += st.con_w9[9 * i + e];
                }
                o.con = put(con.data(), con.size() * sizeof(double));
            }
            bool junction_in = false;
            for (int r = 0; r < st.n_src; ++r) {
                const dnagpu_chain_source& sc = st.src[r];
                if (!sc.m || !sc.k || !sc.pos || 3 * sc.k > sc.m->n_max) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad source");
                std::vector<int32_t> inv(st.n_stn, -1);
                for (size_t a = 0; a < sc.k; ++a) {
                    if (sc.pos[a] >= st.n_stn || inv[sc.pos[a]] >= 0) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: bad source station list");
                    inv[sc.pos[a]] = (int32_t)a;
                }
                o.pos[r] = put(sc.pos, sc.k * sizeof(uint32_t));
                o.inv[r] = put(inv.data(), inv.size() * sizeof(int32_t));
                dnagpu_matrix* m = const_cast<dnagpu_matrix*>(sc.m);
                if (sc.junction)
                    if (int rcj = ensure_jrhs(m)) return rcj;
                d.src[r].F = m->F; d.src[r].np = pad128((uint32_t)(3 * sc.k)); d.src[r].k = (uint32_t)sc.k;
                d.src[r].rhs = sc.junction ? m->jrhs : m->jest;
                d.src[r].jest = sc.junction ? m->jest : nullptr;
                junction_in = junction_in || sc.junction;
            }
            if (mo) {
                plan->out[s] = {nullptr, nj, 0};
                continue;
            }
            if ((junction_in || st.out_junction) && !o.has_est) return fail(ctx, DNAGPU_EINVAL, "chain_plan_create: a junction needs the linearisation point");
            if (st.out_junction)
                if (int rcj = ensure_jrhs(st.out)) return rcj;
            d.outS = st.out->F; d.outnp = pad128(nj);
            d.out_rhs = st.out_junction ? st.out->jrhs : st.out->jest;
            d.out_jest = st.out_junction ? st.out->jest : nullptr;
            plan->out[s] = {st.out, nj, st.out_junction};
        }
    }
    {
        hipError_t e = dnagpu::poison_malloc(&plan->blob, blob.size() + 16);
        if (e == hipSuccess && plan->keeps) e = dnagpu::poison_malloc(&plan->factors, (size_t)factor_bytes + 16);
        if (e == hipSuccess) e = dnagpu::poison_malloc(&plan->table, n_steps * sizeof(CbStep));
        if (e != hipSuccess) {      // (nothing half allocated stays behind: the caller runs the chains step by step and needs the memory for that)
            (void)hipGetLastError();
            return fail(ctx, e == hipErrorOutOfMemory ? DNAGPU_ENOMEM : DNAGPU_EHIP, "chain_plan_create: device allocation", e);
        }
    }
    uint8_t* base = (uint8_t*)plan->blob;
    for (size_t s = 0; s < n_steps; ++s) {
        CbStep& d = table[s];
        const Offs& o = offs[s];
        d.map = o.has_map ? (const int32_t*)(base + o.map) : nullptr;
        d.keep = o.has_map ? (const uint32_t*)(base + o.keep) : nullptr;
        d.xe = o.has_map ? (double*)(base + o.xe) : nullptr;
        d.rhs = o.has_map ? (double*)(base + o.rhs) : nullptr;
        d.est = o.has_est ? (const double* const*)(base + o.est) : nullptr;
        d.con = o.has_con ? (const double*)(base + o.con) : nullptr;
        for (uint32_t r = 0; r < d.n_src; ++r) {
            d.src[r].pos = (const uint32_t*)(base + o.pos[r]);
            d.src[r].inv = (const int32_t*)(base + o.inv[r]);
        }
        d.X = (o.has_map && plan->keeps) ? plan->factors + x_off[s] : nullptr;
        plan->X[s] = d.X;
    }
    HIPCHK(hipMemcpy(plan->blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(plan->table, table.data(), n_steps * sizeof(CbStep), hipMemcpyHostToDevice));
    *out = plan.release();
    return DNAGPU_OK;
}

int dnagpu_chain_plan_run(dnagpu_ctx* ctx, int chain, dnagpu_chain_plan* plan, size_t batch) {
    CHK_CTX();
    CHK_CHAIN();
    if (!plan || batch + 1 >= plan->batch_first.size()) return fail(ctx, DNAGPU_EINVAL, "chain_plan_run: bad arguments");
    const uint32_t first = plan->batch_first[batch], nb = plan->batch_first[batch + 1] - first;
    const dnagpu_chain_plan::Shape sh = plan->shape[batch];
    if (!sh.npp) return fail(ctx, DNAGPU_EINVAL, "chain_plan_run: a batch of kept blocks (dnagpu_partial_complete_factor_planned takes those)");
    // The members' matrices and panels live in scratch of the plan's own size on this chain -- NOT in the chain's batch workspaces, which are
    // as large as the largest BLOCK the chain has batched (a network of large blocks with small junctions: 3 GB per member) --; member 0's
    // panels are the chain's W, as in every batched call.
    int rc = ensure_ws(ctx, chain, sh.npp);
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    const CbStep* table = (const CbStep*)plan->table + first;
    const size_t wspan = ((size_t)batch_panel_cols(sh.nip, sh.njp) + 128) * sh.npp, msq = (size_t)sh.npp * sh.npp;
    const size_t per = msq + wspan + (plan->keeps ? 0 : msq);
    if (ctx->plan_scratch_cap[chain] < (size_t)nb * per) {
        HIPCHK(hipStreamSynchronize(st));
        if (ctx->plan_scratch[chain]) hipFree(ctx->plan_scratch[chain]);
        ctx->plan_scratch[chain] = nullptr;
        ctx->plan_scratch_cap[chain] = 0;
        HIPCHK(dnagpu::poison_malloc(&ctx->plan_scratch[chain], (size_t)nb * per * sizeof(double)));
        ctx->plan_scratch_cap[chain] = (size_t)nb * per;
    }
    CbMembers mem{};
    double* F[BATCH_MAX];
    double* X[BATCH_MAX];
    double* P[BATCH_MAX];
    for (uint32_t b = 0; b < nb; ++b) {
        double* base = ctx->plan_scratch[chain] + (size_t)b * per;
        F[b] = base;
        P[b] = b ? base + msq : ws.W;
        X[b] = plan->keeps ? plan->X[first + b] : base + msq + wspan;
        mem.F[b] = F[b];
        mem.X[b] = X[b];
    }
    launch_cb_rhs(table, nb, st);
    launch_cb_assemble(table, nb, mem, sh.npp, 0, sh.npp, st);
    HIPCHK(hipGetLastError());
    if (nb > 1) {
        InvBatch& bt = ws.batch;
        bt = InvBatch();
        bt.nb = (int)nb;
        bt.add(F[0], msq, F);
        bt.add(X[0], msq, X);
        bt.add(P[0], wspan, P);
    }
    sym_spine_async(ws, F[0], X[0], (int)sh.npp, (int)(sh.nip / 128), (int)(sh.njp / 128));
    ws.batch = InvBatch();
    launch_cb_post(table, nb, mem, sh.nip, sh.npp, sh.outnp_max, st);
    HIPCHK(hipGetLastError());
    for (uint32_t b = 0; b < nb; ++b) {
        const dnagpu_chain_plan::Out& o = plan->out[first + b];
        o.m->n = o.nj;
        o.m->np = pad128(o.nj);
        o.m->form = o.junction ? 1 : 0;
    }
    plan->factored[batch] = plan->keeps ? 1 : 0;
    if (ws.hold_info) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    rc = check_info_batch(ctx, chain, (int)nb, nullptr);
    if (rc) plan->factored[batch] = 0;
    return rc;
}

int dnagpu_chain_plan_info(const dnagpu_chain_plan* plan, int* keeps_factors, double* factor_bytes) {
    if (!plan) return DNAGPU_EINVAL;
    if (keeps_factors) *keeps_factors = plan->keeps ? 1 : 0;
    if (factor_bytes) *factor_bytes = plan->factor_bytes;
    return DNAGPU_OK;
}

int dnagpu_chain_plan_run_rhs(dnagpu_ctx* ctx, int chain, dnagpu_chain_plan* plan, size_t batch_lo, size_t batch_hi) {
    CHK_CTX();
    CHK_CHAIN();
    if (!plan || batch_lo >= batch_hi || batch_hi >= plan->batch_first.size()) return fail(ctx, DNAGPU_EINVAL, "chain_plan_run_rhs: bad arguments");
    for (size_t q = batch_lo; q < batch_hi; ++q)
        if (!plan->factored[q] || !plan->shape[q].npp) return fail(ctx, DNAGPU_EINVAL, "chain_plan_run_rhs: a step has no factor yet");
    const uint32_t first = plan->batch_first[batch_lo], n = plan->batch_first[batch_hi] - first;
    gemm_profile_close(ctx->ws[chain]);
    launch_cb_rhs_steps((const CbStep*)plan->table + first, n, ctx->stream[chain]);
    HIPCHK(hipGetLastError());
    return DNAGPU_OK;
}

int dnagpu_partial_complete_factor_planned(dnagpu_ctx* ctx, int chain, dnagpu_chain_plan* plan, size_t batch, dnagpu_partial* const* pf, int* failed_member) {
    CHK_CTX();
    CHK_CHAIN();
    if (failed_member) *failed_member = -1;
    if (!plan || batch + 1 >= plan->batch_first.size() || plan->shape[batch].npp || !pf) return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor_planned: bad arguments");
    const uint32_t first = plan->batch_first[batch];
    const int nb = (int)(plan->batch_first[batch + 1] - first);
    for (int b = 0; b < nb; ++b) {
        if (!pf[b] || !pf[b]->valid || !pf[b]->spine || pf[b]->nj != plan->out[first + b].nj || pf[b]->nip != pf[0]->nip || pf[b]->njp != pf[0]->njp)
            return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor_planned: bad arguments");
        for (int q = 0; q < b; ++q)
            if (pf[q] == pf[b]) return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor_planned: a member listed twice");
    }
    const uint32_t nip = pf[0]->nip, njp = pf[0]->njp, npp = pf[0]->npp;
    int rc = ensure_batch_ws(ctx, chain, nb, npp, batch_panel_cols(nip, njp));
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    CbMembers mem{};
    for (int b = 0; b < nb; ++b) {
        pf[b]->valid = false;
        pf[b]->completed = false;
        mem.F[b] = b ? ws.bX[b] : ws.X;      // scratch: the kept block's factorisation passes through it
    }
    launch_cb_assemble((const CbStep*)plan->table + first, (uint32_t)nb, mem, npp, nip, njp, st);
    HIPCHK(hipGetLastError());
    set_batch(ws, nb, npp, pf);
    sym_spine_kept_async(ws, ws.X, pf[0]->X, (int)npp, (int)(nip / 128), (int)(njp / 128));
    ws.batch = InvBatch();
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    rc = check_info_batch(ctx, chain, nb, failed_member);
    for (int b = 0; b < nb; ++b) pf[b]->factored = rc == DNAGPU_OK;
    return rc;
}

int dnagpu_chain_reserve(dnagpu_ctx* ctx, int chain, uint32_t n_max) {
    CHK_CTX();
    CHK_CHAIN();
    return ensure_ws(ctx, chain, pad128(n_max) + 256);
}

int dnagpu_copy_stage_reserve(dnagpu_ctx* ctx, int chain, size_t doubles) {
    CHK_CTX();
    CHK_CHAIN();
    int rc = DNAGPU_OK;
    if (doubles) ensure_copy_stage(ctx, chain, doubles, &rc);     // (no room now: the copies take their blocking path when they come)
    return rc;
}

int dnagpu_batch_reserve(dnagpu_ctx* ctx, int chain, uint32_t n_max, uint32_t k_max, int nb_wanted, int* nb_granted) {
    CHK_CTX();
    CHK_CHAIN();
    if (!nb_granted || nb_wanted < 1 || !k_max || k_max > n_max) return fail(ctx, DNAGPU_EINVAL, "batch_reserve: bad arguments");
    nb_wanted = std::min(nb_wanted, (int)BLOCK_BATCH_MAX);
    const uint32_t njp = pad128(k_max + 1), nip = pad128(n_max - k_max ? n_max - k_max : 1), npp = nip + njp;
    *nb_granted = 1;
    if (nb_wanted > 1) {
        int rc = ensure_batch_ws(ctx, chain, nb_wanted, npp, batch_panel_cols(nip, njp));
        if (rc == DNAGPU_OK) {
            *nb_granted = nb_wanted;
            return DNAGPU_OK;
        }
        if (rc != DNAGPU_ENOMEM) return rc;      // (out of memory: nothing of the members' workspaces stays allocated)
    }
    return ensure_ws(ctx, chain, npp);
}

int dnagpu_block_form_reduce_batched(dnagpu_ctx* ctx, int chain, int nb, const uint32_t* blks, const uint32_t* const* con_stn, const double* const* con_w9,
                                     const size_t* n_con, const uint32_t* const* idx_keep, const size_t* k, dnagpu_matrix* const* red,
                                     dnagpu_partial* const* keep, int* failed_member) {
    CHK_CTX();
    CHK_CHAIN();
    if (failed_member) *failed_member = -1;
    if (nb < 1 || nb > BLOCK_BATCH_MAX || !blks || !con_stn || !con_w9 || !n_con || !idx_keep || !k || !red || !keep)
        return fail(ctx, DNAGPU_EINVAL, "block_form_reduce_batched: bad arguments");
    Block* blk[BATCH_MAX];
    uint32_t nip = 0, njp = 0, npp = 0;
    for (int b = 0; b < nb; ++b) {
        blk[b] = find_block(ctx, blks[b]);
        if (!blk[b] || !red[b] || !keep[b] || !keep[b]->spine || !k[b] || !idx_keep[b] || k[b] > blk[b]->n_stn || 3 * k[b] > red[b]->n_max ||
            (n_con[b] && (!con_stn[b] || !con_w9[b])))
            return fail(ctx, DNAGPU_EINVAL, "block_form_reduce_batched: bad arguments");
        const uint32_t n = 3 * blk[b]->n_stn, nj = (uint32_t)(3 * k[b]), ni = n - nj;
        uint32_t nip_b = ni ? pad128(ni) : 0, njp_b = pad128(nj + 1);
        if (ni && keep[b]->n_cap - keep[b]->k_cap >= nip_b && keep[b]->k_cap >= njp_b) {       // (the factor's capacity is the member's shape: schur_eliminate)
            nip_b = keep[b]->n_cap - keep[b]->k_cap;
            njp_b = keep[b]->k_cap;
        }
        if (b == 0) {
            nip = nip_b; njp = njp_b; npp = nip + njp;
        } else if (nip_b != nip || njp_b != njp) {
            return fail(ctx, DNAGPU_EINVAL, "block_form_reduce_batched: the members differ in shape");
        }
        if (npp > keep[b]->n_cap || njp > keep[b]->k_cap) return fail(ctx, DNAGPU_EINVAL, "schur: retained factor capacity");
        for (int q = 0; q < b; ++q)
            if (blks[q] == blks[b] || keep[q] == keep[b] || red[q] == red[b]) return fail(ctx, DNAGPU_EINVAL, "block_form_reduce_batched: a member listed twice");
        for (size_t i = 0; i < n_con[b]; ++i)
            if (con_stn[b][i] >= blk[b]->n_stn) return fail(ctx, DNAGPU_EINVAL, "schur: constraint station out of range");
    }
    if (!nip) return fail(ctx, DNAGPU_EINVAL, "block_form_reduce_batched: nothing to eliminate");
    int rc = ensure_batch_ws(ctx, chain, nb, npp, batch_panel_cols(nip, njp));
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    // every member's normals formed in its elimination order, in its own matrix: one launch per kernel for all members (their lists of
    // constraints must have device copies of their own -- the cache of stage_u32 / stage_f64 --, else one member at a time as before)
    FormBatch fb{};
    bool merged = true;
    for (int b = 0; b < nb; ++b) {
        const uint32_t n = 3 * blk[b]->n_stn, nj = (uint32_t)(3 * k[b]);
        int slot = -1;
        const int32_t* map_dev = nullptr;
        const uint32_t* spos_dev = nullptr;
        rc = schur_order(ctx, blk[b], idx_keep[b], k[b], nip, nj, npp, &slot, &map_dev, &spos_dev);
        if (rc) return rc;
        dnagpu_partial* kp = keep[b];
        kp->valid = false;
        kp->completed = false;
        kp->factored = false;
        kp->n = n; kp->nj = nj; kp->nip = nip; kp->njp = njp; kp->npp = npp;
        if (kp->store) kp->store->n = 0;
        uint32_t* dstn = nullptr;
        double* dw = nullptr;
        if (n_con[b] && merged) {
            rc = stage_u32(ctx, chain, con_stn[b], n_con[b], &dstn);
            if (!rc) rc = stage_f64(ctx, chain, con_w9[b], n_con[b] * 9, &dw);
            if (rc) return rc;
            if (dstn == ctx->scr_u32[chain] || dw == ctx->scr_f64[chain]) merged = false;
        }
        Block* B = blk[b];
        FormMember& m = fb.m[b];
        m.F = b ? ws.bX[b] : ws.X;
        m.map = map_dev; m.map_out = kp->map; m.spos = spos_dev;
        m.prow = B->pair_row; m.pcol = B->pair_col; m.poff = B->pair_off; m.pent = B->pair_ent; m.wblk = B->Wblk;
        m.con_stn = dstn; m.con_w9 = dw; m.rhs = B->rhs[chain];
        m.n_pairs = B->n_pairs; m.n_gnss_blk = B->n_wblk; m.terr_shift = (uint32_t)chain * (B->n_tblk + B->n_dsblk); m.n_con = (uint32_t)n_con[b];
        m.rhs_row = nip + nj;
    }
    if (merged) {
        HbmTimed timed(ctx, chain, DNAGPU_HBM_FORM_ORDERED, 4.0 * (double)npp * npp * nb);
        launch_form_ordered_batch(fb, nb, npp, npp, st);
        HIPCHK(hipGetLastError());
    }
    for (int b = 0; b < nb && !merged; ++b) {
        const uint32_t n = 3 * blk[b]->n_stn, nj = (uint32_t)(3 * k[b]);
        int slot = -1;
        const int32_t* map_dev = nullptr;
        const uint32_t* spos_dev = nullptr;
        rc = schur_order(ctx, blk[b], idx_keep[b], k[b], nip, nj, npp, &slot, &map_dev, &spos_dev);
        if (rc) return rc;
        dnagpu_partial* kp = keep[b];
        kp->valid = false;
        kp->completed = false;
        kp->factored = false;
        kp->n = n; kp->nj = nj; kp->nip = nip; kp->njp = njp; kp->npp = npp;
        if (kp->store) kp->store->n = 0;
        double* F = b ? ws.bX[b] : ws.X;
        uint32_t* dstn = nullptr;
        double* dw = nullptr;
        if (n_con[b]) {
            rc = stage_u32(ctx, chain, con_stn[b], n_con[b], &dstn);
            if (!rc) rc = stage_f64(ctx, chain, con_w9[b], n_con[b] * 9, &dw);
            if (rc) return rc;
        }
        Block* B = blk[b];
        {
            HbmTimed timed(ctx, chain, DNAGPU_HBM_FORM_ORDERED, 4.0 * (double)npp * npp);
            launch_form_ordered(F, npp, npp, map_dev, spos_dev, B->pair_row, B->pair_col, B->pair_off, B->pair_ent, B->Wblk, B->n_pairs, B->n_wblk,
                                (uint32_t)chain * (B->n_tblk + B->n_dsblk), dstn, dw, (uint32_t)n_con[b], B->rhs[chain], nip + nj, st);
        }
        HIPCHK(hipMemcpyAsync(kp->map, map_dev, (size_t)npp * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        if (n_con[b]) HIPCHK(hipStreamSynchronize(st));     // (the staging buffers are the chain's: the next member's lists go through them)
    }
    // the elimination of all members in lock step
    set_batch(ws, nb, npp, keep);
    sym_spine_async(ws, ws.X, keep[0]->X, (int)npp, (int)(nip / 128), (int)(njp / 128));
    ws.batch = InvBatch();
    {
        ExtractBatch eb{};
        uint32_t npj_max = 0;
        for (int b = 0; b < nb; ++b) {
            dnagpu_partial* kp = keep[b];
            const double* F = b ? ws.bX[b] : ws.X;
            kp->valid = true;
            red[b]->n = kp->nj;
            red[b]->np = pad128(kp->nj);
            eb.m[b] = {F + (size_t)nip * npp + nip, kp->X, red[b]->F, red[b]->jest, kp->nj, red[b]->np};
            npj_max = std::max(npj_max, red[b]->np);
        }
        launch_extract_batch(eb, nb, nip, npp, npj_max, st);      // (+ the passenger row's panel entries cleared)
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info_batch(ctx, chain, nb, failed_member);
}

int dnagpu_partial_complete_factor_batched(dnagpu_ctx* ctx, int chain, int nb, dnagpu_partial* const* pf, const dnagpu_matrix* const* kk, int* failed_member) {
    CHK_CTX();
    CHK_CHAIN();
    if (failed_member) *failed_member = -1;
    if (nb < 1 || nb > BLOCK_BATCH_MAX || !pf || !kk) return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor_batched: bad arguments");
    for (int b = 0; b < nb; ++b) {
        if (!pf[b] || !pf[b]->valid || !pf[b]->spine || !kk[b] || kk[b]->n != pf[b]->nj || pf[b]->nip != pf[0]->nip || pf[b]->njp != pf[0]->njp)
            return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor_batched: bad arguments");
        for (int q = 0; q < b; ++q)
            if (pf[q] == pf[b]) return fail(ctx, DNAGPU_EINVAL, "partial_complete_factor_batched: a member listed twice");
    }
    const uint32_t nip = pf[0]->nip, njp = pf[0]->njp, npp = pf[0]->npp;
    int rc = ensure_batch_ws(ctx, chain, nb, npp, batch_panel_cols(nip, njp));
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    for (int b = 0; b < nb; ++b) {
        pf[b]->valid = false;
        pf[b]->completed = false;
        double* F = b ? ws.bX[b] : ws.X;      // scratch: the kept block's factorisation passes through it
        launch_partial_set_trailing(F + (size_t)nip * npp + nip, npp, njp, kk[b]->F, kk[b]->np, pf[b]->nj, st);
    }
    set_batch(ws, nb, npp, pf);
    sym_spine_kept_async(ws, ws.X, pf[0]->X, (int)npp, (int)(nip / 128), (int)(njp / 128));
    ws.batch = InvBatch();
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    rc = check_info_batch(ctx, chain, nb, failed_member);
    for (int b = 0; b < nb; ++b) pf[b]->factored = rc == DNAGPU_OK;
    return rc;
}

int dnagpu_partial_finish_batched(dnagpu_ctx* ctx, int chain, int nb, dnagpu_partial* const* pf, dnagpu_matrix* const* inv) {
    CHK_CTX();
    CHK_CHAIN();
    if (nb < 1 || nb > BLOCK_BATCH_MAX || !pf || !inv) return fail(ctx, DNAGPU_EINVAL, "partial_finish_batched: bad arguments");
    for (int b = 0; b < nb; ++b) {
        if (!pf[b] || !pf[b]->factored || !pf[b]->spine || !inv[b] || pf[b]->n > inv[b]->n_max || pf[b]->nip != pf[0]->nip || pf[b]->njp != pf[0]->njp)
            return fail(ctx, DNAGPU_EINVAL, "partial_finish_batched: bad arguments");
        for (int q = 0; q < b; ++q)
            if (pf[q] == pf[b] || inv[q] == inv[b]) return fail(ctx, DNAGPU_EINVAL, "partial_finish_batched: a member listed twice");
    }
    const uint32_t nip = pf[0]->nip, njp = pf[0]->njp, npp = pf[0]->npp;
    int rc = ensure_batch_ws(ctx, chain, nb, npp, batch_panel_cols(nip, njp));
    if (rc) return rc;
    InvWorkspace& ws = ctx->ws[chain];
    hipStream_t st = ctx->stream[chain];
    gemm_profile_close(ws);
    for (int b = 0; b < nb; ++b) {
        pf[b]->factored = false;
        pf[b]->completed = false;
    }
    set_batch(ws, nb, npp, pf);
    sym_spine_finish_async(ws, ws.X, pf[0]->X, (int)npp, (int)(nip / 128), (int)(njp / 128));
    ws.batch = InvBatch();
    {
        // (padding and un-permutation of all members: two launches)
        UnpermuteBatch ub{};
        double bytes = 0.0;
        for (int b = 0; b < nb; ++b) {
            inv[b]->n = pf[b]->n;
            inv[b]->np = pad128(pf[b]->n);
            ub.m[b] = {b ? ws.bX[b] : ws.X, pf[b]->map, inv[b]->F, inv[b]->n, inv[b]->np};
            bytes += 16.0 * (double)inv[b]->np * inv[b]->np;
        }
        HbmTimed t(ctx, chain, DNAGPU_HBM_UNPERMUTE, bytes);
        launch_unpermute_batch(ub, nb, npp, st);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemsetAsync(ws.info, 0x7f, BATCH_MAX * sizeof(int), st));
    HIPCHK(hipMemcpyAsync(ws.info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_info(ctx, chain);
}

int dnagpu_block_load_reduced(dnagpu_ctx* ctx, int chain, uint32_t rblk, uint32_t src_blk, const uint32_t* idx_keep, size_t k,
                              const dnagpu_matrix* red, dnagpu_matrix* m) {
    CHK_CTX();
    CHK_CHAIN();
    Block* rb = find_block(ctx, rblk);
    Block* sb = find_block(ctx, src_blk);
    if (!rb || !sb || !red || !idx_keep || rb->n_stn != k || red->n != 3 * k || (m && red->n > m->n_max))
        return fail(ctx, DNAGPU_EINVAL, "block_load_reduced: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (idx_keep[i] >= sb->n_stn) return fail(ctx, DNAGPU_EINVAL, "block_load_reduced: station out of range");
    hipStream_t st = ctx->stream[chain];
    uint32_t* didx = nullptr;
    int rc = stage_u32(ctx, chain, idx_keep, k, &didx);
    if (rc) return rc;
    if (m) {      // (m = NULL: right-hand side and linearisation point only -- a step whose factor is kept, dnagpu_schur_carry_rhs)
        m->n = red->n;
        m->np = red->np;
        HIPCHK(hipMemcpyAsync(m->F, red->F, (size_t)red->np * red->np * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(rb->rhs[chain], red->jest, (size_t)red->n * sizeof(double), hipMemcpyDeviceToDevice, st));
    launch_gather_vec3(sb->x_orig, didx, (uint32_t)k, rb->x_est[chain], st);
    return DNAGPU_OK;
}

int dnagpu_junction_scatter(dnagpu_ctx* ctx, int chain, dnagpu_matrix* dst, const uint32_t* idx_to, size_t k, const dnagpu_matrix* jm) {
    CHK_CTX();
    CHK_CHAIN();
    if (!dst || !jm || (k && !idx_to) || jm->n != 3 * k) return fail(ctx, DNAGPU_EINVAL, "junction_scatter: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (3 * (uint64_t)idx_to[i] + 2 >= dst->n) return fail(ctx, DNAGPU_EINVAL, "junction_scatter: station out of range");
    if (!k) return DNAGPU_OK;
    uint32_t* didx = nullptr;
    int rc = stage_u32(ctx, chain, idx_to, k, &didx);
    if (rc) return rc;
    launch_junction_scatter(dst->F, dst->np, didx, (uint32_t)k, jm->F, jm->np, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_junction_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk_to, const uint32_t* idx_to, size_t k, const dnagpu_matrix* jm) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk_to);
    if (!b || !jm || (k && !idx_to) || jm->n != 3 * k) return fail(ctx, DNAGPU_EINVAL, "junction_rhs: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (idx_to[i] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "junction_rhs: station out of range");
    if (!k) return DNAGPU_OK;
    uint32_t* didx = nullptr;
    int rc = stage_u32(ctx, chain, idx_to, k, &didx);
    if (rc) return rc;
    launch_junction_rhs(b->rhs[chain], b->x_est[chain], didx, (uint32_t)k, jm->F, jm->np, jm->jest, jm->form == 1 ? jm->jrhs : nullptr,
                        ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_block_add_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, const uint32_t* idx, size_t k, const dnagpu_matrix* jm, int zero_first) {
    CHK_CTX();
    CHK_CHAIN();
    Block* b = find_block(ctx, blk);
    if (!b || !jm || (k && !idx) || jm->n != 3 * k) return fail(ctx, DNAGPU_EINVAL, "block_add_rhs: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (idx[i] >= b->n_stn) return fail(ctx, DNAGPU_EINVAL, "block_add_rhs: station out of range");
    if (zero_first) HIPCHK(hipMemsetAsync(b->rhs[chain], 0, (size_t)3 * b->n_stn * sizeof(double), ctx->stream[chain]));
    if (!k) return DNAGPU_OK;
    uint32_t* didx = nullptr;
    int rc = stage_u32(ctx, chain, idx, k, &didx);
    if (rc) return rc;
    launch_scatter_add_vec3(b->rhs[chain], didx, (uint32_t)k, jm->jest, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_block_gather_stations(dnagpu_ctx* ctx, int chain, uint32_t dst_blk, const uint32_t* dst_pos, uint32_t src_blk, const uint32_t* src_idx,
                                 size_t k) {
    CHK_CTX();
    CHK_CHAIN();
    Block* d = find_block(ctx, dst_blk);
    Block* sb = find_block(ctx, src_blk);
    if (!d || !sb || (k && (!dst_pos || !src_idx))) return fail(ctx, DNAGPU_EINVAL, "block_gather_stations: bad arguments");
    for (size_t i = 0; i < k; ++i)
        if (dst_pos[i] >= d->n_stn || src_idx[i] >= sb->n_stn) return fail(ctx, DNAGPU_EINVAL, "block_gather_stations: station out of range");
    if (!k) return DNAGPU_OK;
    std::vector<uint32_t> both(dst_pos, dst_pos + k);
    both.insert(both.end(), src_idx, src_idx + k);
    uint32_t* didx = nullptr;
    int rc = stage_u32(ctx, chain, both.data(), 2 * k, &didx);
    if (rc) return rc;
    launch_copy_vec3_indexed(d->x_est[chain], didx, sb->x_orig, didx + k, (uint32_t)k, ctx->stream[chain]);
    return DNAGPU_OK;
}

int dnagpu_junction_get_estimates(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* jm, double* est) {
    CHK_CTX();
    CHK_CHAIN();
    if (!jm || (!est && jm->n)) return fail(ctx, DNAGPU_EINVAL, "junction_get_estimates: bad arguments");
    return d2h(ctx, chain, est, jm->jest, (size_t)jm->n * sizeof(double));
}

int dnagpu_junction_put_estimates(dnagpu_ctx* ctx, int chain, dnagpu_matrix* jm, const double* est, size_t k) {
    CHK_CTX();
    CHK_CHAIN();
    if (!jm || 3 * k > jm->n_max || (k && !est)) return fail(ctx, DNAGPU_EINVAL, "junction_put_estimates: bad arguments");
    jm->form = 0;        // (estimates from outside are adjusted estimates: the estimates form, whatever the matrix held before)
    if (!k) return DNAGPU_OK;
    HIPCHK(hipMemcpyAsync(jm->jest, est, 3 * k * sizeof(double), hipMemcpyHostToDevice, ctx->stream[chain]));
    HIPCHK(hipStreamSynchronize(ctx->stream[chain]));
    return DNAGPU_OK;
}

}  // extern "C"

/* ---- diagnostics ------------------------------------------------------------------
 * Times `reps` launches of one tile-GEMM variant on scratch data (values irrelevant).
 * variant: 0 = NT, 1 = NN, 2 = TN.  Returns average ms per launch and the flops of one launch. */
extern "C" int dnagpu_bench_gemm(dnagpu_ctx* ctx, int variant, int mt, int nt, int K, int kmode, int lower, int reps, double* avg_ms,
                                 double* flops) {
    CHK_CTX();
    if (mt <= 0 || nt <= 0 || K <= 0 || K % 16 || reps <= 0) return fail(ctx, DNAGPU_EINVAL, "bench_gemm: bad arguments");
    size_t M = (size_t)mt * 128, N = (size_t)nt * 128;
    size_t ld = std::max(std::max(M, N), (size_t)K);
    size_t cols = ld;
    double *A = nullptr, *B = nullptr, *Cc = nullptr;
    HIPCHK(dnagpu::poison_malloc(&A, ld * cols * sizeof(double)));
    HIPCHK(dnagpu::poison_malloc(&B, ld * cols * sizeof(double)));
    HIPCHK(dnagpu::poison_malloc(&Cc, ld * cols * sizeof(double)));
    // pseudo-random fill (full-range mantissas: zero fill would flatter the clocks)
    std::vector<double> h(ld * 1024);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = (double)(int64_t)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    for (size_t off = 0; off < ld * cols; off += h.size()) {
        size_t cnt = std::min(h.size(), ld * cols - off);
        hipMemcpy(A + off, h.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
        hipMemcpy(B + off, h.data() + 7, (cnt - 7) * sizeof(double), hipMemcpyHostToDevice);
    }
    hipMemset(Cc, 0, ld * cols * sizeof(double));
    GemmArgs a;
    a.A = A; a.B = B; a.C = Cc; a.lda = a.ldb = a.ldc = (int)ld;
    a.mt = mt; a.nt = nt; a.K = K; a.alpha = 1.0; a.beta = 0.0; a.kmode = kmode; a.lower = lower; a.mirror = 0;
    int akc = variant == 2, bkc = variant >= 1;
    hipStream_t st = ctx->stream[0];
    HIPCHK(gemm_attach_order(ctx->ws[0], a));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch_gemm(a, akc, bkc, st);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) launch_gemm(a, akc, bkc, st);
    hipEventRecord(e1, st);
    hipError_t e = hipStreamSynchronize(st);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(A); hipFree(B); hipFree(Cc);
    if (e != hipSuccess) return fail(ctx, DNAGPU_EHIP, "bench_gemm", e);
    if (avg_ms) *avg_ms = ms / reps;
    if (flops) {
        double f = 0.0;
        for (int it = 0; it < mt; ++it) {
            int jmax = lower ? it : nt - 1;
            for (int jt = 0; jt <= jmax; ++jt) {
                int kb = 0, ke = K;
                if (kmode == KM_LE_J) ke = (jt + 1) * 128;
                if (kmode == KM_GE_J) kb = jt * 128;
                if (kmode == KM_LE_I) ke = (it + 1) * 128;
                if (kmode == KM_GE_I) kb = it * 128;
                if (ke > K) ke = K;
                if (ke > kb) f += 2.0 * 128 * 128 * (ke - kb);
            }
        }
        *flops = f;
    }
    return DNAGPU_OK;
}
