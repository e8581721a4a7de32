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
#include <cstdint>
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
    struct Super {
        double work;
        int si, sj;
    };
    std::vector<Super> supers;
    int smt = (mt + G - 1) / G, snt = (nt + G - 1) / G;
    for (int si = 0; si < smt; ++si)
        for (int sj = 0; sj < snt; ++sj) {
            double w = 0.0;
            int cnt = 0;
            for (int it = si * G; it < std::min(mt, (si + 1) * G); ++it)
                for (int jt = sj * G; jt < std::min(nt, (sj + 1) * G); ++jt) {
                    if (lower && jt > it) continue;
                    if (jt < jlo || jt >= jhi) continue;
                    w += klen(it, jt, K, kmode, tile) + 16;  // + fixed per-tile cost
                    ++cnt;
                }
            if (cnt) supers.push_back({w, si, sj});
        }
    std::stable_sort(supers.begin(), supers.end(), [](const Super& a, const Super& b) { return a.work > b.work; });
    std::vector<uint32_t> lists[8];
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Super& s : supers) {
        int x = 0;
        for (int c = 1; c < 8; ++c)
            if (load[c] < load[x]) x = c;
        load[x] += s.work;
        // inside a super-tile: column-major over the tile patch, heaviest tiles first is
        // not needed (the patch runs concurrently on the XCD's 32 CUs)
        for (int jt = s.sj * G; jt < std::min(nt, (s.sj + 1) * G); ++jt)
            for (int it = s.si * G; it < std::min(mt, (s.si + 1) * G); ++it) {
                if (lower && jt > it) continue;
                if (jt < jlo || jt >= jhi) continue;
                lists[x].push_back(((uint32_t)it << 16) | (uint32_t)jt);
            }
    }
    size_t maxlen = 0;
    for (auto& l : lists) maxlen = std::max(maxlen, l.size());
    out.assign(maxlen * 8, 0xffffffffu);
    for (int x = 0; x < 8; ++x)
        for (size_t q = 0; q < lists[x].size(); ++q) out[q * 8 + x] = lists[x][q];
    return out;
}

}  // namespace dnagpu
