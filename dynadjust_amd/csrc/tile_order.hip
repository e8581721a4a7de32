// Workgroup -> tile assignment for the tile GEMM.
//
// MI355X dispatches workgroup b to XCD b % 8 and each XCD has its own 4 MiB L2.  The
// triangular operands of the inverse (k <= i, k >= i, ...) make the work per tile very
// uneven, so a plain "contiguous chunk per XCD" mapping leaves most XCDs idle while one
// finishes (measured: 29-34 TFLOP/s instead of 70).  The table built here
//   1. cuts the tile grid into G x G super-tiles (operand panels shared inside an L2),
//   2. sorts the super-tiles by work, heaviest first,
//   3. deals them to the 8 XCDs greedily (least-loaded XCD gets the next super-tile),
//   4. interleaves the 8 per-XCD lists so that entry b belongs to XCD b % 8.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "la_kernels.h"

namespace dnagpu {

// it / jt count tiles of `tile` columns; the k restrictions are in units of the recursion's 128-blocks
static inline int klen(int it, int jt, int K, int kmode, int tile) {
    int kb = 0, ke = K;
    const int bi = it * tile / 128, bj = jt * tile / 128;
    switch (kmode) {
        case KM_LE_J: ke = (bj + 1) * 128; break;
        case KM_GE_J: kb = bj * 128; break;
        case KM_LE_I: ke = (bi + 1) * 128; break;
        case KM_GE_I: kb = bi * 128; break;
        default: break;
    }
    if (ke > K) ke = K;
    return ke > kb ? ke - kb : 0;
}

// Columns (in 128-tiles) [lo[q], lo[q + 1]) of a launch for each of `world` ranks, balanced by the work of their tiles: the
// intra-block distributed inverse gives every rank the tile columns of one range (sym_inverse.hip)
std::vector<int> split_tile_columns(int mt128, int nt128, int K, int kmode, int lower, int world) {
    std::vector<double> w(nt128, 0.0);
    double total = 0.0;
    for (int jt = 0; jt < nt128; ++jt) {
        for (int it = lower ? jt : 0; it < mt128; ++it) w[jt] += klen(it, jt, K, kmode, 128) + 16;
        total += w[jt];
    }
    std::vector<int> lo(world + 1, nt128);
    lo[0] = 0;
    double acc = 0.0;
    int q = 1;
    for (int jt = 0; jt < nt128 && q < world; ++jt) {
        acc += w[jt];
        while (q < world && acc >= total * q / world) lo[q++] = jt + 1;
    }
    for (; q < world; ++q) lo[q] = nt128;
    return lo;
}

// WHO SHARES WHAT, AND FOR HOW LONG.  The workgroups of a patch share operand panels through their XCD's L2 only while they are at
// the same k.  With full k ranges they start together, run at the same rate and finish together, so do their successors: 79 % L2
// hits (87.5 % would be perfect for 8 x 8).  With a triangular operand the k range of a tile depends on its row (k <= i, k >= i) or
// its column (k <= j, k >= j): the patch starts together at the k all ranges share, but its rows (columns) finish one tile of k
// after the other, their slots are refilled at different times, and the successors never meet: measured 18 % hits when the tiles
// that replace a finished row come from a column of the next patch, 42-50 % when they come from one row (then at least these 8
// run in step for good).  Hence:
//   * list order inside a patch = tiles of equal length together (rows when the length depends on i, columns when on j);
//     MEASURED (round 2, profiles/r02_tile_order_l2.txt, n = 19 968): rows together took the LAUUM from 19 % to 48 % hits, 140 -> 91 GB
//     of fabric reads and 70.3 -> 73.9 TFLOP/s (at 3.7 TB/s the reads WERE the bound); k <= i 18 % -> 31 %, 67.3 -> 69.4.
//     (Walking the tiles of two patches in PAIRS -- one away from the common end of the ranges, its partner towards it -- raised the hits to
//     60-74 % and made nothing faster, twice: below ~3 TB/s the kernel no longer waits for the fabric.  Removed in round 5;
//     profiles/HISTORY.md.)
std::vector<uint32_t> build_tile_order(int mt128, int nt128, int K, int kmode, int lower, int tile, int jt_lo128, int jt_hi128) {
    std::vector<uint32_t> out;
    const int mt = mt128 * (128 / tile), nt = nt128 * (128 / tile);
    const int jlo = jt_lo128 < 0 ? 0 : jt_lo128 * (128 / tile), jhi = jt_hi128 < 0 ? nt : jt_hi128 * (128 / tile);
    long total = 0;
    for (int it = 0; it < mt; ++it)
        for (int jt = jlo; jt < jhi && jt <= (lower ? it : nt - 1); ++jt) ++total;
    if (total <= 0) return out;
    if (total <= 8) {
        for (int it = 0; it < mt; ++it)
            for (int jt = jlo; jt < jhi && jt <= (lower ? it : nt - 1); ++jt) out.push_back(((uint32_t)it << 16) | (uint32_t)jt);
        return out;
    }
    int T = std::max(mt, nt);
    int G = T >= 64 ? 8 : T >= 32 ? 4 : T >= 16 ? 2 : 1;
    // Short k ranges: a tile lives a few hundred microseconds, its operand panels are small, and what counts is that the eight XCDs
    // finish together -- patches of 2 x 2 deal the work finer (measured on the elimination's launches, K = 1 500 ... 3 600: 62.7 ->
    // 63.6 TFLOP/s, cfg3 one chain 3.17 -> 3.10 s).  Long ranges keep the 8 x 8 patches: there the fabric reads are the risk (the
    // LAUUM, K = n).
    if (K < 6144) G = std::min(G, 2);
    const bool by_rows = kmode == KM_LE_I || kmode == KM_GE_I;
    constexpr uint32_t NONE = 0xffffffffu;

    struct Tile {
        int it, jt, cls;
    };
    struct Patch {
        double work = 0.0;
        int si = 0, sj = 0;
        std::vector<Tile> tiles;      // tiles of equal length adjacent
    };
    std::vector<Patch> patches;
    int smt = (mt + G - 1) / G, snt = (nt + G - 1) / G;
    for (int si = 0; si < smt; ++si)
        for (int sj = 0; sj < snt; ++sj) {
            Patch p;
            p.si = si;
            p.sj = sj;
            const int i_lo = si * G, i_hi = std::min(mt, (si + 1) * G), j_lo = sj * G, j_hi = std::min(nt, (sj + 1) * G);
            auto push = [&](int it, int jt) {
                if (lower && jt > it) return;
                if (jt < jlo || jt >= jhi) return;
                p.tiles.push_back({it, jt, by_rows ? it - i_lo : jt - j_lo});
                p.work += klen(it, jt, K, kmode, tile) + 16;  // + fixed per-tile cost
            };
            if (by_rows) {
                for (int it = i_lo; it < i_hi; ++it)
                    for (int jt = j_lo; jt < j_hi; ++jt) push(it, jt);
            } else {
                for (int jt = j_lo; jt < j_hi; ++jt)
                    for (int it = i_lo; it < i_hi; ++it) push(it, jt);
            }
            if (!p.tiles.empty()) patches.push_back(std::move(p));
        }

    // units = what is dealt to an XCD as a whole: a patch
    struct Unit {
        double work = 0.0;
        std::vector<uint32_t> e;
    };
    std::vector<Unit> units;
    for (const Patch& p : patches) {
        Unit u;
        u.work = p.work;
        for (const Tile& t : p.tiles) u.e.push_back(((uint32_t)t.it << 16) | (uint32_t)t.jt);
        units.push_back(std::move(u));
    }
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.work > b.work; });
    std::vector<uint32_t> lists[8];
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Unit& u : units) {
        int x = 0;
        for (int c = 1; c < 8; ++c)
            if (load[c] < load[x]) x = c;
        load[x] += u.work;
        lists[x].insert(lists[x].end(), u.e.begin(), u.e.end());
    }
    size_t maxlen = 0;
    for (auto& l : lists) maxlen = std::max(maxlen, l.size());
    out.assign(maxlen * 8, NONE);
    for (int x = 0; x < 8; ++x)
        for (size_t q = 0; q < lists[x].size(); ++q) out[q * 8 + x] = lists[x][q];
    return out;
}

}  // namespace dnagpu
