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

// Tiles-per-launch from which the tiles of the triangular k ranges are walked in pairs (below); 0 = never, which is the default:
// measured, the pairs raise the L2 hit rates as intended and do not make anything faster (see the end of the comment below).
// DNAGPU_PAIR_TILES / dnagpu_debug_set_pair_tiles set it.
static std::atomic<long> g_pair_tiles{getenv("DNAGPU_PAIR_TILES") ? atol(getenv("DNAGPU_PAIR_TILES")) : 0};
long pair_tiles_get() { return g_pair_tiles.load(); }
long pair_tiles_set(long tiles) { return g_pair_tiles.exchange(tiles < 0 ? 0 : tiles); }
static long pair_threshold() { return g_pair_tiles.load(); }

// WHO SHARES WHAT, AND FOR HOW LONG.  The workgroups of a patch share operand panels through their XCD's L2 only while they are at
// the same k.  With full k ranges they start together, run at the same rate and finish together, so do their successors: 79 % L2
// hits (87.5 % would be perfect for 8 x 8).  With a triangular operand the k range of a tile depends on its row (k <= i, k >= i) or
// its column (k <= j, k >= j): the patch starts together at the k all ranges share, but its rows (columns) finish one tile of k
// after the other, their slots are refilled at different times, and the successors never meet: measured 18 % hits when the tiles
// that replace a finished row come from a column of the next patch, 42-50 % when they come from one row (then at least these 8
// run in step for good).  Hence:
//   * list order inside a patch = tiles of equal length together (rows when the length depends on i, columns when on j);
//   * opt-in, large launches: PAIRS.  A workgroup computes a tile of class l (row or column offset in its patch) of patch P, walking k
//     away from the common end, and then the tile of class G-1-l of a patch Q, walking k TOWARDS the common end.  The first
//     phase ends staggered, one class after the other -- and exactly that stagger puts the second phase in step: the class that
//     starts first in Q has the longest range, and every later starter enters at the k the earlier ones have just reached.  All
//     workgroups of the pair finish together (length(l) + length(G-1-l) is the same for all l), so the next pair starts
//     together.  Which patches walk which way is a function of their position alone (odd class index = towards), so a tile is
//     summed in the same order in every table that contains it (the split tables of the distributed inverse included).
//     MEASURED (round 2, profiles/r02_tile_order_l2.txt, n = 19 968): rows together took the LAUUM from 19 % to 48 % hits, 140 -> 91 GB
//     of fabric reads and 70.3 -> 73.9 TFLOP/s (at 3.7 TB/s the reads WERE the bound); k <= i 18 % -> 31 %, 67.3 -> 69.4.  Pairs on top:
//     hits 60-74 %, reads down another 25-50 % -- and the launches 0-4 % SLOWER (LAUUM 71.6, inverse 126.4 -> 130.3 ms): below
//     ~3 TB/s the kernel no longer waits for the fabric, and workgroups twice as long pack worse into the last wave.
std::vector<uint32_t> build_tile_order(int mt128, int nt128, int K, int kmode, int lower, int tile, int jt_lo128, int jt_hi128, int* pairs_out) {
    std::vector<uint32_t> out;
    if (pairs_out) *pairs_out = 0;
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
    static const int g_env = getenv("DNAGPU_PATCH") ? atoi(getenv("DNAGPU_PATCH")) : 0;      // probe: patch edge for every launch
    static const int k_short = getenv("DNAGPU_PATCH_SHORT_K") ? atoi(getenv("DNAGPU_PATCH_SHORT_K")) : 6144;
    if (K < k_short) G = std::min(G, 2);
    if (g_env > 0) G = g_env;
    static const int row_major_env = getenv("DNAGPU_TILE_ROWS") ? atoi(getenv("DNAGPU_TILE_ROWS")) : -1;     // probe: 0 columns always, 1 rows always
    const bool by_rows = row_major_env >= 0 ? row_major_env != 0 : (kmode == KM_LE_I || kmode == KM_GE_I);
    // the decision to pair is taken on the WHOLE launch (not on a rank's column range), so that every table of a shape agrees
    const long whole = lower ? (long)mt128 * (mt128 + 1) / 2 : (long)mt128 * nt128;
    const bool alt = tile == 128 && G >= 4 && kmode != KM_FULL && pair_threshold() > 0 && whole >= pair_threshold() && gemm_128_takes_pairs();
    constexpr uint32_t NONE = 0xffffffffu, FLIP = 0x8000u;

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

    // units = what is dealt to an XCD as a whole: a patch, or a pair of patches; entries = one or two tiles per workgroup
    struct Unit {
        double work = 0.0;
        std::vector<uint32_t> e;
    };
    std::vector<Unit> units;
    const int width = alt ? 2 : 1;
    auto code = [&](const Tile& t, bool flip) { return ((uint32_t)t.it << 16) | (uint32_t)t.jt | (flip ? FLIP : 0u); };
    if (!alt) {
        for (const Patch& p : patches) {
            Unit u;
            u.work = p.work;
            for (const Tile& t : p.tiles) u.e.push_back(code(t, false));
            units.push_back(std::move(u));
        }
    } else {
        std::vector<const Patch*> away, towards;
        for (const Patch& p : patches) (((by_rows ? p.si : p.sj) & 1) ? towards : away).push_back(&p);
        auto heavier = [](const Patch* a, const Patch* b) {
            if (a->tiles.size() != b->tiles.size()) return a->tiles.size() > b->tiles.size();     // like with like: whole patches first
            return a->work > b->work;
        };
        std::stable_sort(away.begin(), away.end(), heavier);
        std::stable_sort(towards.begin(), towards.end(), heavier);
        // the range towards the common end is the longer the HIGHER the class for k <= ..., the LOWER the class for k >= ...
        const bool towards_long_first_is_high = (kmode == KM_LE_I || kmode == KM_LE_J);
        auto by_class = [](const Patch& p) {
            std::vector<std::vector<Tile>> c;
            int last = -1;
            for (const Tile& t : p.tiles) {
                if (t.cls != last) c.emplace_back();
                last = t.cls;
                c.back().push_back(t);
            }
            return c;      // ascending class (the patch lists its tiles that way)
        };
        const size_t np = std::max(away.size(), towards.size());
        for (size_t q = 0; q < np; ++q) {
            Unit u;
            std::vector<std::vector<Tile>> ca, cb;
            if (q < away.size()) {
                ca = by_class(*away[q]);
                u.work += away[q]->work;
            }
            if (q < towards.size()) {
                cb = by_class(*towards[q]);
                u.work += towards[q]->work;
            }
            const int na = (int)ca.size(), nb = (int)cb.size(), c = std::max(na, nb) - 1;
            std::vector<char> used(nb, 0);
            for (int r = 0; r < na; ++r) {
                const int rb = c - r;
                std::vector<Tile>* lb = (rb >= 0 && rb < nb) ? &cb[rb] : nullptr;
                if (lb) used[rb] = 1;
                for (size_t x = 0; x < ca[r].size(); ++x) {
                    u.e.push_back(code(ca[r][x], false));
                    u.e.push_back(lb && x < lb->size() ? code((*lb)[x], true) : NONE);
                }
                if (lb)
                    for (size_t x = ca[r].size(); x < lb->size(); ++x) {       // more tiles in Q's class than in P's: on their own
                        u.e.push_back(code((*lb)[x], true));
                        u.e.push_back(NONE);
                    }
            }
            // classes of Q nobody is paired with: the longest range first (they then enter in step like a second phase would)
            for (int k = 0; k < nb; ++k) {
                const int rb = towards_long_first_is_high ? nb - 1 - k : k;
                if (used[rb]) continue;
                for (const Tile& t : cb[rb]) {
                    u.e.push_back(code(t, true));
                    u.e.push_back(NONE);
                }
            }
            units.push_back(std::move(u));
        }
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
    for (auto& l : lists) maxlen = std::max(maxlen, l.size() / width);
    out.assign(maxlen * 8 * width, NONE);
    for (int x = 0; x < 8; ++x)
        for (size_t q = 0; q < lists[x].size() / width; ++q)
            for (int w = 0; w < width; ++w) out[(q * 8 + x) * width + w] = lists[x][q * width + w];
    if (pairs_out) *pairs_out = alt ? 1 : 0;
    return out;
}

}  // namespace dnagpu
