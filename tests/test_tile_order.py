"""The workgroup -> tile tables of the tile GEMM (dynadjust_amd/csrc/tile_order.hip), computed on the host: every tile of a launch exactly
once whatever the shape, the k restriction and a rank's column range; entry b belongs to XCD b % 8; the column ranges of a split launch
(intra-block distributed inverse) together cover the whole launch."""
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
    assert per.value == 1
    return np.frombuffer(buf, dtype=np.uint32, count=n).copy()


def decode(e):
    return int(e >> 16), int(e & 0xFFFF)


@pytest.mark.parametrize("mt,nt,lower,kmode", [
    (156, 156, 1, "k>=i"), (78, 78, 0, "k<=j"), (78, 78, 0, "k>=j"), (78, 78, 0, "k<=i"), (78, 78, 1, "full"), (117, 118, 0, "k<=i"),
    (39, 39, 0, "k>=j"), (40, 39, 0, "k<=j"), (33, 33, 1, "k>=i"), (20, 19, 0, "k<=i"), (9, 9, 1, "k>=i"), (3, 2, 0, "full"), (70, 70, 1, "k>=i")])
def test_every_tile_once(built, mt, nt, lower, kmode):
    km = KM[kmode]
    K = max(mt, nt) * 128
    tab = table(built, mt, nt, K, km, lower)
    whole = mt * (mt + 1) // 2 if lower else mt * nt
    assert len(tab) % 8 == 0 or len(tab) <= 8
    seen = set()
    for e in tab:
        if e == NONE:
            continue
        it, jt = decode(e)
        assert 0 <= it < mt and 0 <= jt < nt and (not lower or jt <= it)
        assert (it, jt) not in seen
        seen.add((it, jt))
    assert len(seen) == whole


@pytest.mark.parametrize("world", [2, 3, 4])
def test_split_tables_agree_with_the_whole(built, world):
    """the column ranges of a split launch (intra-block distributed inverse): together every tile of the unsplit table, each once"""
    mt = nt = 117
    K = mt * 128
    for km, lower in ((4, 1), (1, 0), (3, 0)):
        ref = {decode(e) for e in table(built, mt, nt, K, km, lower) if e != NONE}
        got = set()
        bounds = [round(q * nt / world) for q in range(world + 1)]
        for q in range(world):
            for e in table(built, mt, nt, K, km, lower, 128, bounds[q], bounds[q + 1]):
                if e == NONE:
                    continue
                it, jt = decode(e)
                assert bounds[q] <= jt < bounds[q + 1] and (it, jt) not in got
                got.add((it, jt))
        assert got == ref


def test_small_launch_tables(built):
    """64- and 32-tile tables (small and tiny launches): every sub-tile once"""
    for tile, f in ((64, 2), (32, 4)):
        tab = table(built, 10, 10, 1280, 4, 1, tile=tile)
        tiles = [decode(e) for e in tab if e != NONE]
        assert len(tiles) == len(set(tiles)) == (10 * f) * (10 * f + 1) // 2
