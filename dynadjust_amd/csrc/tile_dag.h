// Tile-DAG executor: a whole blocked factorisation / inverse as ONE kernel launch.
//
// The recursion of sym_inverse.hip issues its work as a sequence of products (tile-GEMM launches) and leaves (potrf + trtri of a
// 128 x 128 diagonal tile), every one waiting for the one before: at n = 20 000 an elimination is ~1 600 dependent launches, the
// large ones each paying a cold first wave and a partly filled last one, the small ones bound by launch latency, and nothing of
// step s + 1 can start before the last tile of step s has ended (round 2: 49.5 TFLOP/s for the elimination against 74 for the tile
// kernel alone).  Here the same sequence is RECORDED instead (DagBuilder), cut into its tile tasks, the dependencies between the
// tasks are derived from the tiles they read and write (read-after-write, write-after-read, write-after-write, at 128 x 128
// granularity -- exact, whatever the recursion does), and one launch runs them all:
//   * the tasks form ONE list, ordered by a list-scheduling simulation with critical-path priorities (the leading block of step
//     s + 1 is factored while the trailing update of step s is still under way -- look-ahead --, a product's last tiles run beside
//     the next one's first);
//   * a launch is `workers` persistent workgroups.  A worker takes the next position of the list (an atomic ticket), waits until the
//     completion flags of that task's predecessors carry this launch's epoch, acquires, runs the task with the very tile code of
//     the per-product kernels (gemm_tile_dma.h, gemm_tile_reg.h, leaf_body.h: same bits), releases, raises the task's own flag
//     (a plain store: nobody but its successors' workers looks at it), and goes back for the next ticket;
//   * the list is a topological order, so a waiting worker only ever waits for tasks whose tickets were taken before its own --
//     by workers that are running: no deadlock, whatever the dispatch order and however few workers are resident;
//   * `workers` bounds what a launch can hold of the GPU: chains that run their graphs side by side share the slots by agreement
//     (512 / chains each) instead of one chain's waiting workers starving the others.
// Two other schedulers were built and measured first (round 3, profiles/r03_dag_schedulers.txt): one workgroup per task in list
// order (25 - 35 % of the slot time went to workgroups waiting on a task while later ones were ready -- with one chain there was
// nothing else to run, with four the waiting workgroups took the others' slots), and a dataflow scheduler (predecessor counters,
// ready queues by priority, direct hand-off to waiting workgroups: no slot is ever wasted, but every hop on the critical path pays
// four dependent device-scope read-modify-writes, 13 us median, and the factorisation is critical-path bound: 2.3 x slower).
// Replaces the dpotrf / dpotri call pair of matrix_2d::cholesky_inverse (dynadjust/include/math/dnamatrix_contiguous.cpp:982-1006).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <memory>
#include <vector>
#include "la_kernels.h"

namespace dnagpu {

constexpr int DAG_MAX_BUFS = 4;

// products: operand layouts NT (akc = 0, bkc = 0), NN (akc = 0, bkc = 1), TN (akc = 1, bkc = 1); + DAG_TILE64: a 64 x 64 tile by the
// register-staged latency shape (gemm_tile_reg.h) instead of a 128 x 128 tile by the LDS-DMA throughput shape (gemm_tile_dma.h) -- the
// products with few tiles, which sit on the critical path: four workgroups share what one would do, each in a quarter of the time
enum DagTaskType : uint8_t { DAG_GEMM_NT = 0, DAG_GEMM_NN = 1, DAG_GEMM_TN = 2, DAG_LEAF = 3, DAG_TILE64 = 4 };
enum DagTaskFlags : uint8_t { DAG_ALPHA_NEG = 1, DAG_BETA_ONE = 2, DAG_MIRROR = 4, DAG_DOWN = 8 };

struct DagTask {
    uint32_t a_off, b_off, c_off;   // element offsets of the product's operands in their buffers (leaf: a = the tile in the matrix, c = the tile in X)
    uint16_t it, jt;                // the task's tile of C relative to the product's C, in tiles of its own size (128 or 64)
    uint16_t kb, ke;                // its k range in 128-tiles (leaf: kb = the tile's position on the diagonal, for `info`)
    uint8_t type, bufs, flags, pad; // DagTaskType; buffer numbers a | b << 2 | c << 4; DagTaskFlags
    uint32_t id;                    // its number in the recorded order = the index of its completion flag
    uint32_t dep0, ndep;            // its predecessors: deps[dep0 .. dep0 + ndep), runs of recorded numbers
};
static_assert(sizeof(DagTask) == 36, "DagTask layout");

// `count` tasks with the recorded numbers first, first + stride, ...
struct DagRun {
    uint32_t first;
    uint16_t count, stride;
};

struct DagGraph {
    // host copies (kept: the self-test executes them on the CPU)
    std::vector<DagTask> tasks;     // in launch order (a topological order)
    std::vector<DagRun> deps;
    uint32_t nids = 0;
    double flops = 0.0;             // of the tile products (2 * tile^2 * k per task), as issued
    uint32_t n_products = 0, n_leaves = 0;
    double sim_makespan_us = 0.0, sim_work_us = 0.0, critical_path_us = 0.0;   // of the list-scheduling simulation (duration model)
    // device copies
    int device = -1;
    DagTask* d_tasks = nullptr;
    DagRun* d_deps = nullptr;
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
    // reorder: 0 = recorded order, 1 = list-scheduling order for `workers` concurrent workgroups
    std::shared_ptr<DagGraph> finish(int reorder, int workers);

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
    const DagRun* deps;
    uint32_t ntasks;
    uint32_t epoch;
    uint32_t* flags;
    unsigned long long* ticket;     // device word, only ever incremented; `ticket_base` = its value when this launch starts
    unsigned long long ticket_base;
    int paranoid;                   // diagnostic (DNAGPU_DAG_PARANOID): every wave acquires again after the barrier
    int workers;                    // workgroups of the launch (each takes tickets until they run out: ntasks + workers in all)
    double* buf0; double* buf1; double* buf2; double* buf3;
    int ld0, ld1, ld2, ld3;
    int* info;
    unsigned long long* trace;      // diagnostic (DNAGPU_DAG_TRACE): per task 4 words -- wall clock (100 MHz) with the ticket, with the predecessors done, at the end; when its own work was done
};
void launch_tile_dag(const DagLaunch& L, hipStream_t s);

// uploads g's arrays to the current device (once)
hipError_t dag_upload(DagGraph& g);

// CPU execution of a graph on host buffers (tests: the dependency analysis must make every order the flags admit give the bits of
// the recorded order).  order: 0 = the list's own order (must be a topological order of its own flags), 1 = random among the ready
// tasks, 2 = always the LAST ready task of the list (the most out-of-order execution the flags admit).  Returns false on a stall.
bool dag_execute_host(const DagGraph& g, double* const buf[DAG_MAX_BUFS], const int ld[DAG_MAX_BUFS], int order, uint64_t seed);

}  // namespace dnagpu
