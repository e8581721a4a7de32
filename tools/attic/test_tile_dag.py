"""The tile-DAG path (csrc/tile_dag.h; opt-in, DNAGPU_DAG=1): a whole factorisation / inverse recorded as tile tasks and run as
one launch of persistent workgroups.

CPU: the dependency analysis.  The recorded sequence of every driver (inverse, elimination, completion, the light kept factor) is
executed on host buffers in its recorded order and -- from the same inputs -- in the launch order, a random admissible order and
the most out-of-order one the completion flags admit: all must give the recorded order's bits.  With the write-after-read
dependencies withheld (a hook in the builder) the same check must FAIL, i.e. it can see a missing dependency.

GPU: the DAG launch against the per-product path on the same matrices: bit-identical (the tasks are the per-product kernels' own
tile code), repeatedly (a race would show as an occasional difference); and an adjustment end to end against the oracle.
Replaces dpotrf / dpotri inside matrix_2d::cholesky_inverse (dynadjust/include/math/dnamatrix_contiguous.cpp:982-1006)."""
import ctypes as C
import os

import numpy as np
import pytest

from dynadjust_amd import _lib

KINDS = {"inverse": 1, "schur": 2, "schur_keep": 3, "complete": 4, "spine": 5, "spine_kept": 6, "spine_finish": 7}


def _selftest(lib, kind, ti, tj, what=0, seed=7):
    st = np.zeros(6)
    rc = lib.dnagpu_debug_tile_dag_selftest(KINDS[kind], ti, tj, what, seed, st.ctypes.data_as(_lib.c_f64p))
    return rc, st


@pytest.mark.parametrize("kind,ti,tj,what", [
    ("inverse", 1, 0, 0), ("inverse", 2, 0, 0), ("inverse", 5, 0, 0), ("schur", 4, 2, 0), ("schur", 1, 1, 0), ("schur_keep", 4, 2, 0),
    ("complete", 3, 2, 3), ("complete", 3, 2, 1), ("complete", 3, 2, 2), ("spine", 5, 2, 0), ("spine_kept", 4, 3, 0), ("spine_finish", 5, 2, 0),
    ("spine", 9, 1, 0),
])
def test_every_admissible_order_gives_the_recorded_bits(built, kind, ti, tj, what):
    rc, st = _selftest(built, kind, ti, tj, what)
    assert rc == 0, (rc, st)
    assert st[0] >= 1 and st[4] > 0 and st[3] >= st[4] * 0.999          # tasks; critical path <= simulated makespan


def test_mixed_task_sizes(built):
    """products of at least 8 tiles as 128 x 128 tasks, the smaller ones as 64 x 64 tasks: a tile written by four small tasks and read by a
    large one (and the other way round) -- what the full-size graphs consist of"""
    old = built.dnagpu_debug_set_small_tiles(8)
    try:
        for kind, ti, tj, what in (("inverse", 7, 0, 0), ("spine", 8, 2, 0), ("spine_finish", 8, 1, 0), ("complete", 5, 2, 3)):
            rc, st = _selftest(built, kind, ti, tj, what, seed=11)
            assert rc == 0, (kind, rc)
    finally:
        built.dnagpu_debug_set_small_tiles(old)


def test_the_check_notices_a_missing_dependency(built):
    """without the write-after-read dependencies the block columns of the finished factor are overwritten while products still read
    them (sym_inverse.hip spine_finish): some admissible order must then differ from the recorded one"""
    os.environ["DNAGPU_DAG_TEST_DROP_WAR"] = "1"
    try:
        rc, _ = _selftest(built, "spine_finish", 5, 2)
    finally:
        del os.environ["DNAGPU_DAG_TEST_DROP_WAR"]
    assert rc > 0
    rc, _ = _selftest(built, "spine_finish", 5, 2)
    assert rc == 0


def _spd_packed(n, rng):
    a = rng.standard_normal((n, 24))
    m = a @ a.T + np.diag(np.linspace(1.0, 1e3, n))
    return np.concatenate([m[j:, j] for j in range(n)])          # packed lower, column-major (matrix_2d::packed_index)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [300, 1536, 4200])
def test_dag_launch_equals_the_per_product_path(built, n):
    from dynadjust_amd.device import DeviceContext
    rng = np.random.default_rng(n)
    ap = _spd_packed(n, rng)
    ns = n // 3
    with DeviceContext(0) as ctx:
        m = ctx.matrix(n)
        ctx.block_create(0, ns, 0)
        ctx.block_set_stations(0, np.zeros(3 * ns))
        keep = np.arange(ns - max(1, ns // 50), ns, dtype=np.uint32)
        red = ctx.matrix(3 * len(keep))
        inv = ctx.matrix(n)
        pf = ctx.partial_create(n, 3 * len(keep))
        old = built.dnagpu_debug_set_tile_dag(0)
        try:
            def run(what):
                m.upload_packed(ap, n)
                if what == "inverse":
                    m.invert()
                    return m.download_packed()
                if what == "eliminate":
                    ctx.block_reduce(0, m, keep, red)
                    return red.download_packed()
                ctx.block_reduce(0, m, keep, red, keep=pf)
                ctx.partial_complete(pf, red, inv, n)
                return inv.download_packed()
            ref = {w: run(w) for w in ("inverse", "eliminate", "keep")}
            l0, t0 = C.c_uint64(), C.c_uint64()
            built.dnagpu_tile_dag_stats(ctx.h, C.byref(l0), C.byref(t0))
            built.dnagpu_debug_set_tile_dag(1)
            for rep in range(6):
                for w in ("inverse", "eliminate", "keep"):
                    assert np.array_equal(run(w), ref[w]), (w, rep)
            l1, t1 = C.c_uint64(), C.c_uint64()
            built.dnagpu_tile_dag_stats(ctx.h, C.byref(l1), C.byref(t1))
            assert l1.value - l0.value >= (18 if n > 256 else 0) and t1.value > t0.value      # the launches really were DAG launches
        finally:
            built.dnagpu_debug_set_tile_dag(old)
        ctx.partial_destroy(pf)
        for q in (m, red, inv):
            q.close()
        ctx.block_destroy(0)


@pytest.mark.gpu
@pytest.mark.parametrize("mt", [False, True])
def test_adjustment_on_the_dag_path_against_the_oracle(built, orc, tmp_path, mt):
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "n", 36, 30, 0, 4, seed=21)
    net = orc.Network(str(tmp_path / "n"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    old = built.dnagpu_debug_set_tile_dag(1)
    try:
        a = adjust.DnaAdjust()
        a.PrepareAdjustment(adjust.ProjectSettings("n", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=mt))
        assert a.AdjustNetwork() == ost and a.CurrentIteration() == o.iterations()
        l, t = C.c_uint64(), C.c_uint64()
        built.dnagpu_tile_dag_stats(a.device_context(), C.byref(l), C.byref(t))
        assert l.value > 0
        for k in range(4):
            assert np.abs(a.block_estimates(k) - o.block_estimates(k)).max() < 1e-8
            vo = o.block_variances(k)
            assert np.abs(a.block_variances_packed(k) - vo).max() / np.abs(vo).max() < 1e-8
        a.close()
    finally:
        built.dnagpu_debug_set_tile_dag(old)
    o.close()
