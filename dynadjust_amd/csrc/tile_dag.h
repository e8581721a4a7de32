// Tile-DAG executor: a whole blocked factorisation / inverse as ONE kernel launch.
//
// The recursion of sym_inverse.hip issues its work as a sequence of products (tile-GEMM launches) and leaves (potrf + trtri of a
// 128 x 128 diagonal tile), every one waiting for the one before: at n = 20 000 an elimination is ~1 600 dependent launches, the
// large ones each paying a cold first wave and a partly filled last one, the small ones bound by launch latency, and nothing of
// step s + 1 can start before the last tile of step s has ended (round 2: 49.5 TFLOP/s for the elimination against 74 for the tile
// kernel alone).  Here the same sequence is RECORDED instead (DagBuilder), cut into its tile tasks, the dependencies between the
// tasks are derived from the tiles they read and write (read-after-write, write-after-read, write-after-write, at 128 x 128
// granularity -- exact, whatever the recursion does), and one launch runs them all as a dataflow graph:
//   * every task carries a counter of unfinished predecessors; a task that finishes releases its results (agent scope), decrements
//     the counters of its successors, and whichever decrement reaches zero puts that successor into a ready queue;
//   * one workgroup per task: a workgroup takes ONE task out of the ready queues (waiting while they are empty), acquires, runs it
//     with the very tile code of the per-product kernels (gemm_tile_dma.h, gemm_tile_reg.h, leaf_body.h: same bits), and ends.  A
//     workgroup never sits on a task that is not ready while another one is, and a slot that frees is re-arbitrated between all
//     kernels on the GPU, so several chains' graphs share the chip tile by tile;
//   * sixteen ready queues by critical-path length (longest remaining path first): the leading block of step s + 1 is factored
//     while the trailing update of step s is still under way (look-ahead), and the last tiles of a product run beside the first
//     ones of the next;
//   * no deadlock by construction: a workgroup only ever waits for a queue entry, and as long as tasks remain, one of them has all
//     its predecessors finished or running.
// Replaces the dpotrf / dpotri call pair of matrix_2d::cholesky_inverse (dynadjust/include/math/dnamatrix_contiguous.cpp:982-1006).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <memory>
#include <vector>
#include "la_kernels.h"

namespace dnagpu {

constexpr int DAG_MAX_BUFS = 4;
constexpr int DAG_QUEUES = 16;
// words of the state image in front of the predecessor counters: queue heads, queue tails, queue bases (+ end), balance, waiters, handoffs
constexpr uint32_t DAG_STATE_BALANCE = 3 * DAG_QUEUES + 1, DAG_STATE_WAITERS = DAG_STATE_BALANCE + 1, DAG_STATE_HANDOFFS = DAG_STATE_BALANCE + 2;
constexpr uint32_t DAG_STATE_FIXED = DAG_STATE_BALANCE + 3;

// products: operand layouts NT (akc = 0, bkc = 0), NN (akc = 0, bkc = 1), TN (akc = 1, bkc = 1); + DAG_TILE64: a 64 x 64 tile by the
// register-staged latency shape (gemm_tile_reg.h) instead of a 128 x 128 tile by the LDS-DMA throughput shape (gemm_tile_dma.h) -- the
// products with few tiles, which sit on the critical path: four workgroups share what one would do, each in a quarter of the time
enum DagTaskType : uint8_t { DAG_GEMM_NT = 0, DAG_GEMM_NN = 1, DAG_GEMM_TN = 2, DAG_LEAF = 3, DAG_TILE64 = 4 };
enum DagTaskFlags : uint8_t { DAG_ALPHA_NEG = 1, DAG_BETA_ONE = 2, DAG_MIRROR = 4, DAG_DOWN = 8 };

struct DagTask {
    uint32_t a_off, b_off, c_off;   // element offsets of the product's operands in their buffers (leaf: a = the tile in the matrix, c = the tile in X)
    uint16_t it, jt;                // the task's tile of C relative to the product's C, in tiles of its own size (128 or 64)
    uint16_t kb, ke;                // its k range in 128-tiles (leaf: kb = the tile's position on the diagonal, for `info`)
    uint8_t type, bufs, flags;      // DagTaskType; buffer numbers a | b << 2 | c << 4; DagTaskFlags
    uint8_t queue;                  // the ready queue it goes to (0 = most urgent)
    uint32_t id;                    // its number in the recorded order (index of its predecessor counter)
    uint32_t succ0, nsucc;          // its successors: succ[succ0 .. succ0 + nsucc), runs of recorded numbers
};
static_assert(sizeof(DagTask) == 36, "DagTask layout");

// `count` tasks with the recorded numbers first, first + stride, ...
struct DagRun {
    uint32_t first;
    uint16_t count, stride;
};

// What a launch changes is one array of 32-bit words per chain, restored from the graph's image (state_init) before every launch:
//   words [0, Q)                        queue heads: entries taken so far
//   words [Q, 2 Q)                      queue tails: entries put so far
//   words [2 Q, 3 Q]                    slot_base (constant)
//   word  DAG_STATE_BALANCE             tasks put into the queues minus workgroups arrived (signed)
//   words DAG_STATE_WAITERS / HANDOFFS  workgroups that found nothing and wait / tasks handed straight to a waiting workgroup
//   words [state_pending, + nids)       unfinished predecessors per recorded number
//   words [state_slots, + ntasks)       the queues' slots (task + 1, 0 = not yet filled), queue q at slot_base[q]
//   words [state_mail, + ntasks)        mailbox of the w-th waiting workgroup (task + 1)
struct DagGraph {
    // host copies (kept: the self-test executes them on the CPU)
    std::vector<DagTask> tasks;     // sorted by queue, inside a queue by remaining path: queue q owns tasks[slot_base[q] .. slot_base[q + 1])
    std::vector<DagRun> succ;
    std::vector<uint32_t> id2task;  // recorded number -> index in `tasks` (0xffffffff: no such task)
    std::vector<uint32_t> state_init;
    uint32_t slot_base[DAG_QUEUES + 1] = {};
    uint32_t nids = 0, state_pending = 0, state_slots = 0, state_mail = 0;
    double flops = 0.0;             // of the tile products (2 * tile^2 * k per task), as issued
    uint32_t n_products = 0, n_leaves = 0;
    double sim_makespan_us = 0.0, sim_work_us = 0.0, critical_path_us = 0.0;   // list-scheduling simulation with the duration model (diagnostic)
    // device copies
    int device = -1;
    DagTask* d_tasks = nullptr;
    DagRun* d_succ = nullptr;
    uint32_t* d_id2task = nullptr;
    uint32_t* d_state_init = nullptr;
    ~DagGraph();
};

// Records a sequence of products and leaves on symbolic buffers and turns it into a DagGraph.
class DagBuilder {
public:
    // ld[b]: leading dimension (elements, a multiple of 128) of buffer b; products of fewer than `small_tiles` 128-tiles are cut
    // into 64 x 64 tasks (la_kernels.h SMALL_LAUNCH_TILES)
    DagBuilder(int nbuf, const int* ld, long small_tiles);
    double* base(int b) const;      // the (fake) address the recording pass uses for buffer b
    void add_gemm(const GemmArgs& a, int akc, int bkc);
    void add_leaf(const double* A_tile, double* X_tile, int diag_tile);
    // workers: concurrent workgroups the diagnostic simulation assumes
    std::shared_ptr<DagGraph> finish(int workers);

private:
    struct Op {
        uint8_t type, bufs, flags;
        uint32_t a_off, b_off, c_off;
        int a_rt, a_ct, b_rt, b_ct, c_rt, c_ct;     // tile coordinates of the operands' origins in their buffers
        int mt, nt, kt, kmode, lower;
        uint32_t id_base;
    };
    int nbuf_;
    long small_tiles_;
    int ld_[DAG_MAX_BUFS];
    std::vector<Op> ops_;
    uint32_t nids_ = 0;
    bool decode(const void* p, int& buf, uint32_t& off, int& rt, int& ct) const;
};

struct DagLaunch {
    const DagTask* tasks;
    const DagRun* succ;
    const uint32_t* id2task;
    uint32_t* state;                // this chain's copy of the graph's state image
    uint32_t ntasks, state_pending, state_slots, state_mail;
    uint32_t slot_base[DAG_QUEUES + 1];
    double* buf0; double* buf1; double* buf2; double* buf3;
    int ld0, ld1, ld2, ld3;
    int* info;
    unsigned long long* trace;      // diagnostic (DNAGPU_DAG_TRACE): per task 4 words -- wall clock (100 MHz) at start, with its task in hand, at the end; when its own work was done
};
void launch_tile_dag(const DagLaunch& L, hipStream_t s);

// uploads g's arrays to the current device (once)
hipError_t dag_upload(DagGraph& g);

// CPU execution of a graph on host buffers with the device's own protocol (predecessor counters, successor runs, ready queues):
// the dependency analysis must make every order the counters admit give the bits of the recorded order.
//   order: 0 = the recorded order (must never meet a task whose counter is not zero), 1 = a random ready task, 2 = always the ready
//          task recorded LAST (the most out-of-order execution the counters admit), 3 = the queues' own order (most urgent queue
//          first, first in first out)
// Returns false on a stall (tasks left, none ready), a counter that goes below zero or a queue that overflows.
bool dag_execute_host(const DagGraph& g, double* const buf[DAG_MAX_BUFS], const int ld[DAG_MAX_BUFS], int order, uint64_t seed);

}  // namespace dnagpu
