"""Look-ahead inside a factorisation (include/dnagpu.h dnagpu_debug_set_lookahead): the tile columns of a trailing update that the next
diagonal block does not touch run on a side stream while the chain's own stream factors that block.  The hazard tracking must keep the
sequential result, bit for bit (replaces nothing in the reference: dpotrf / dpotri of matrix_2d::cholesky_inverse,
dnamatrix_contiguous.cpp:982-1006, are sequential calls)."""
import ctypes as C

import numpy as np
import pytest

from dynadjust_amd import adjust
from dynadjust_amd.device import pack_lower

pytestmark = pytest.mark.gpu


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n + 8))
    M = A @ A.T / n
    M[np.diag_indices(n)] += 1.0
    return M


def _side_launches(ctx, lib):
    n = C.c_uint64()
    assert lib.dnagpu_lookahead_stats(ctx.h, C.byref(n)) == 0
    return n.value


@pytest.mark.parametrize("n", [1500, 3100])
def test_the_inverse_has_the_same_bits_with_and_without_look_ahead(gpu_ctx, built, n):
    ap = pack_lower(_spd(n, n))
    old = built.dnagpu_debug_set_lookahead(0, -1)
    try:
        ref = gpu_ctx.cholesky_inverse_packed(ap.copy(), n)
        before = _side_launches(gpu_ctx, built)
        built.dnagpu_debug_set_lookahead(1, 1)            # every trailing update with two or more tile columns is split
        got = gpu_ctx.cholesky_inverse_packed(ap.copy(), n)
        assert _side_launches(gpu_ctx, built) > before
        assert np.array_equal(ref, got)
        # and again (tables cached, events reused)
        assert np.array_equal(ref, gpu_ctx.cholesky_inverse_packed(ap.copy(), n))
    finally:
        built.dnagpu_debug_set_lookahead(old, 1024)


@pytest.mark.parametrize("mt,batch", [(False, 16), (True, 16), (True, 0)])
def test_a_phased_adjustment_has_the_same_bits_with_and_without_look_ahead(built, tmp_path, mt, batch):
    adjust.write_synthetic_network(str(tmp_path), "l", 60, 50, 0, 5, seed=33)

    def run():
        a = adjust.DnaAdjust()
        a.PrepareAdjustment(adjust.ProjectSettings("l", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=mt, batch_blocks=batch))
        assert a.AdjustNetwork() == 0
        out = ([a.block_estimates(b) for b in range(a.blockCount())], [a.block_variances_packed(b) for b in range(a.blockCount())])
        a.close()
        return out

    old = built.dnagpu_debug_set_lookahead(0, -1)
    try:
        x0, v0 = run()
        built.dnagpu_debug_set_lookahead(1, 1)
        x1, v1 = run()
    finally:
        built.dnagpu_debug_set_lookahead(old, 1024)
    for b in range(len(x0)):
        assert np.array_equal(x0[b], x1[b]) and np.array_equal(v0[b], v1[b])
