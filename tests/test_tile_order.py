"""The workgroup -> tile tables of the tile GEMM (dynadjust_amd/csrc/tile_order.hip), computed on the host: every tile of a launch exactly
once whatever the shape, the k restriction and a rank's column range; in paired tables the two tiles of a workgroup have complementary
lengths (what keeps the workgroups of a patch at the same k, see the comment there); the walk direction of a tile is the same in every
table that contains it."""
import ctypes as C

import numpy as np
import pytest

KM = {"full": 0, "k<=j": 1, "k>=j": 2, "k<=i": 3, "k>=i": 4}
NONE = 0xFFFFFFFF


def table(lib, mt, nt, K, kmode, lower, tile=128, jlo=-1, jhi=-1):
    per = C.c_int()
    n = lib.dnagpu_debug_tile_order(mt, nt, K, kmode, lower, tile, jlo, jhi, None, 0, C.byref(per))
    buf = (C.c_uint32 * max(n, 1))()
    assert lib.dnagpu_debug_tile_order(mt, nt, K, kmode, lower, tile, jlo, jhi, buf, n, C.byref(per)) == n
    return np.frombuffer(buf, dtype=np.uint32, count=n).copy(), per.value


@pytest.fixture
def paired(built):
    old = built.dnagpu_debug_set_pair_tiles(1024)      # the opt-in pairing of large triangular launches
    yield built
    built.dnagpu_debug_set_pair_tiles(old)


def decode(e):
    return int(e >> 16), int(e & 0x7FFF), bool(e & 0x8000)


def klen(it, jt, K, kmode):
    kb, ke = 0, K
    if kmode == 1: ke = (jt + 1) * 128
    if kmode == 2: kb = jt * 128
    if kmode == 3: ke = (it + 1) * 128
    if kmode == 4: kb = it * 128
    return max(min(ke, K) - kb, 0)


@pytest.mark.parametrize("mt,nt,lower,kmode", [
    (156, 156, 1, "k>=i"), (78, 78, 0, "k<=j"), (78, 78, 0, "k>=j"), (78, 78, 0, "k<=i"), (78, 78, 1, "full"), (117, 118, 0, "k<=i"),
    (39, 39, 0, "k>=j"), (40, 39, 0, "k<=j"), (33, 33, 1, "k>=i"), (20, 19, 0, "k<=i"), (9, 9, 1, "k>=i"), (3, 2, 0, "full"), (70, 70, 1, "k>=i")])
def test_every_tile_once_and_pairs_complementary(paired, mt, nt, lower, kmode):
    built = paired
    km = KM[kmode]
    K = max(mt, nt) * 128
    tab, per = table(built, mt, nt, K, km, lower)
    whole = mt * (mt + 1) // 2 if lower else mt * nt
    # (launches with a short k range are dealt in 2 x 2 patches and never paired)
    assert per == (2 if (km != 0 and whole >= 1024 and max(mt, nt) >= 32 and K >= 6144) else 1)
    assert len(tab) % (8 * per) == 0 or len(tab) <= 8
    seen = {}
    for e in tab:
        if e == NONE:
            continue
        it, jt, flip = decode(e)
        assert 0 <= it < mt and 0 <= jt < nt and (not lower or jt <= it)
        assert (it, jt) not in seen
        seen[(it, jt)] = flip
    assert len(seen) == whole
    if per == 2:
        G = 8 if max(mt, nt) >= 64 else 4
        by_rows = km in (3, 4)
        sums = {}
        npairs = 0
        for w in range(len(tab) // 2):
            e1, e2 = tab[2 * w], tab[2 * w + 1]
            if e1 == NONE:
                assert e2 == NONE
                continue
            i1, j1, f1 = decode(e1)
            assert f1 == bool(((i1 if by_rows else j1) // G) & 1)          # direction = parity of the patch's class index
            if e2 == NONE:
                continue
            i2, j2, f2 = decode(e2)
            assert not f1 and f2                                           # away first, then towards
            npairs += 1
            # complementary classes: all pairs of the same two patches have the same total length
            key = (i1 // G, j1 // G, i2 // G, j2 // G)
            sums.setdefault(key, set()).add(klen(i1, j1, K, km) + klen(i2, j2, K, km))
        assert npairs > 0.35 * whole                                       # most tiles travel in pairs
        assert all(len(v) == 1 for v in sums.values()), [v for v in sums.values() if len(v) > 1][:3]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_split_tables_agree_with_the_whole(paired, world):
    built = paired
    """the column ranges of a split launch (intra-block distributed inverse): together every tile once, each with the direction it has
    in the unsplit table (same summation order -> bit-identical results on one and on several GPUs)"""
    mt = nt = 117
    K = mt * 128
    for km, lower in ((4, 1), (1, 0), (3, 0)):
        whole, _ = table(built, mt, nt, K, km, lower)
        ref = {decode(e)[:2]: decode(e)[2] for e in whole if e != NONE}
        got = {}
        bounds = [round(q * nt / world) for q in range(world + 1)]
        for q in range(world):
            part, _ = table(built, mt, nt, K, km, lower, 128, bounds[q], bounds[q + 1])
            for e in part:
                if e == NONE:
                    continue
                it, jt, flip = decode(e)
                assert bounds[q] <= jt < bounds[q + 1] and (it, jt) not in got
                got[(it, jt)] = flip
        assert got == ref


def test_default_tables_have_no_pairs(built):
    for km, lower in ((4, 1), (1, 0), (3, 0), (0, 1)):
        tab, per = table(built, 156, 156, 156 * 128, km, lower)
        tiles = [decode(e) for e in tab if e != NONE]
        assert per == 1 and len(set(t[:2] for t in tiles)) == len(tiles) == (156 * 157 // 2 if lower else 156 * 156)
        assert not any(f for _, _, f in tiles)


def test_small_launch_tables(built):
    """64-tile tables (small launches) carry no direction bits and no pairs"""
    tab, per = table(built, 10, 10, 1280, 4, 1, tile=64)
    assert per == 1
    tiles = [decode(e) for e in tab if e != NONE]
    assert len(tiles) == 20 * 21 // 2 and not any(f for _, _, f in tiles)
