"""a.batch_blocks: blocks of one shape through the large steps of the condensed schedule as ONE batch of merged launches
(include/dnagpu.h dnagpu_*_batched; replaces the reference's Solve() calls following each other, dnaadjust.cpp:2812 / 3512 / 3556).
Every member must come out with the bits of the unbatched calls."""
import os

import numpy as np
import pytest

from dynadjust_amd import adjust

pytestmark = pytest.mark.gpu


def _run(folder, name, **kw):
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, **kw)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    st = a.AdjustNetwork()
    return a, st


def _results(a):
    B = a.blockCount()
    return ([a.block_estimates(b) for b in range(B)], [a.block_variances_packed(b) for b in range(B)],
            [a.GetIterationCorrection(i + 1) for i in range(a.CurrentIteration())])


@pytest.mark.parametrize("mt", [False, True])
@pytest.mark.parametrize("rows,cols,blocks", [(48, 40, 6), (60, 90, 5)])
def test_batched_blocks_have_the_bits_of_unbatched_ones(built, tmp_path, mt, rows, cols, blocks):
    """strips of equal height: the interior blocks share a shape and go through condensing, rigorous solve and variance matrices
    as one batch; estimates, variances and corrections are bit-identical to the run with batching off"""
    adjust.write_synthetic_network(str(tmp_path), "b", rows, cols, 0, blocks, seed=21)
    a0, st0 = _run(str(tmp_path), "b", multi_thread=mt, batch_blocks=0)
    assert st0 == 0 and a0.batched_block_steps() == 0
    x0, v0, c0 = _results(a0)
    it0 = a0.CurrentIteration()
    a0.close()
    a1, st1 = _run(str(tmp_path), "b", multi_thread=mt, batch_blocks=16)
    assert st1 == 0 and a1.CurrentIteration() == it0
    # at least the interior blocks were batched in every phase: condensing and kept-block factorisation once (a.reuse_factors: the later
    # iterations of this GNSS-only network keep the factors of the first), the variance matrices once
    assert a1.batched_block_steps() >= 2 * 3 and a1.factor_reuses() == (it0 - 1) * blocks
    a2, st2 = _run(str(tmp_path), "b", multi_thread=mt, batch_blocks=16, reuse_factors=False)
    # ... of every iteration when every iteration factors again
    assert st2 == 0 and a2.batched_block_steps() >= 2 * (2 * it0 + 1) and a2.factor_reuses() == 0
    a2.close()
    x1, v1, c1 = _results(a1)
    assert c0 == c1
    for b in range(blocks):
        assert np.array_equal(x0[b], x1[b])
        assert np.array_equal(v0[b], v1[b])
    a1.close()


def test_small_blocks_of_unequal_size_share_a_bucket(built, orc, tmp_path):
    """a dnasegment-like cut (strips of 7 ... 12 rows of 100 stations: ~12 blocks of n = 2 400 ... 3 900, no two alike): blocks whose eliminated
    part falls into one bucket (AssignBatchShapes: an eighth of their size wide) are padded with an identity up to the bucket's largest
    member and go through the batched calls together.  Against the oracle (1e-8 m, 1e-8 relative) and against the run without batching
    (own shapes: the padded ones differ in blocking, so equal to rounding, not bit for bit)."""
    info = adjust.write_synthetic_network(str(tmp_path), "s", 120, 100, 0, 1, seed=9, rows_lo=7, rows_hi=12)
    assert info["blocks"] >= 10
    # (the faster of the host's LAPACKs: on the pool's non-Intel hosts the MKL runtime is five times slower than the OpenBLAS of the scipy wheel,
    #  and this test was three minutes of the suite with it; the whole small-block workload against the oracle: tests/golden/smallblocks_oracle.npz)
    fast = orc.scipy_openblas_path()
    if not (fast and orc.use_lapack(fast)):
        orc.use_mkl(True)
    try:
        orc.load().orc_set_threads(min(os.cpu_count() or 1, 16))
        net = orc.Network(str(tmp_path / "s"), True)
        o = orc.Adjustment(net, True)
        o.prepare()
        ost = o.run()
    finally:
        orc.use_lapack(None)
    a0, st0 = _run(str(tmp_path), "s", multi_thread=True, batch_blocks=0)
    assert st0 == ost and a0.batched_block_steps() == 0
    x0, v0, c0 = _results(a0)
    a0.close()
    a1, st1 = _run(str(tmp_path), "s", multi_thread=True, batch_blocks=16)
    assert st1 == ost and a1.CurrentIteration() == o.iterations()
    B = a1.blockCount()
    sizes = {a1.block_estimates(b).size for b in range(B)}
    assert len(sizes) >= 4                                   # (the blocks really differ)
    assert a1.batched_block_steps() >= B                     # ... and most of them were batched all the same
    assert a1.batched_flops() > 0.5 * a1.algorithmic_flops()
    x1, v1, c1 = _results(a1)
    for b in range(B):
        vo = o.block_variances(b)
        scale = np.abs(vo).max()
        assert np.abs(x1[b] - o.block_estimates(b)).max() < 1e-8 and np.abs(x0[b] - o.block_estimates(b)).max() < 1e-8
        assert np.abs(v1[b] - vo).max() / scale < 1e-8 and np.abs(v0[b] - vo).max() / scale < 1e-8
        assert np.abs(x1[b] - x0[b]).max() < 1e-9 and np.abs(v1[b] - v0[b]).max() / scale < 1e-10
    for i in range(o.iterations()):
        assert abs(c1[i] - o.max_correction(i + 1)) < 1e-8
    a1.close()
    o.close()


@pytest.mark.parametrize("stage", [False, True, "hbm"])
def test_blocks_without_a_kept_factor_make_it_again(built, orc, tmp_path, monkeypatch, stage):
    """the HBM budget denies every block a kept factor (DNAGPU_FACTOR_BUDGET_GB=0: what cfg4 on one GPU does to 121 of its 128 blocks): in a
    GNSS-only network the factor is then formed and eliminated again in the rigorous solve and once more for the variance matrix (chain-owned
    storage) instead of an inverse per iteration.  Same bits as the run in which every block keeps its factor; against the oracle as well."""
    adjust.write_synthetic_network(str(tmp_path), "t", 48, 40, 0, 6, seed=33)
    net = orc.Network(str(tmp_path / "t"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    monkeypatch.delenv("DNAGPU_FACTOR_BUDGET_GB", raising=False)
    a0, st0 = _run(str(tmp_path), "t", multi_thread=True, batch_blocks=0, stage=bool(stage))
    assert st0 == ost and a0.memory_plan()["factors_made_again"] == 0 and a0.memory_plan()["blocks_keeping_their_factor"] == 6
    x0, v0, c0 = _results(a0)
    a0.close()
    monkeypatch.setenv("DNAGPU_FACTOR_BUDGET_GB", "0")
    # staged: the slot of a block's packed variance matrix -- page-locked host memory (round 6), or HBM where the staged store's host part
    # is full ("hbm": DNAGPU_HOST_STORE_GB=0) -- holds the block's packed factor during the iterations: copied back and unpacked where the
    # blocks of the resident run eliminate again
    packed = bool(stage)
    if stage == "hbm":
        monkeypatch.setenv("DNAGPU_HOST_STORE_GB", "0")
    a1, st1 = _run(str(tmp_path), "t", multi_thread=True, stage=bool(stage))
    plan = a1.memory_plan()
    assert st1 == ost and plan["blocks_keeping_their_factor"] == 0 and plan["blocks_without_kept_factor_refactor"]
    if packed:
        assert plan["blocks_packing_their_factor"] == 6 and plan["factors_made_again"] == 0
        assert plan["factors_taken_from_their_packed_copy"] == 6 * (a1.CurrentIteration() + 1)
        if stage == "hbm":
            assert plan["staged_variances_host_bytes"] == 0 and plan["staged_variances_packed_in_hbm_bytes"] > 0
        else:
            assert plan["staged_variances_host_bytes"] > 0 and plan["staged_variances_packed_in_hbm_bytes"] == 0
    else:
        assert plan["factors_made_again"] == 6 * (a1.CurrentIteration() + 1)          # every rigorous solve + every variance matrix
    x1, v1, c1 = _results(a1)
    assert c0 == c1
    for b in range(6):
        assert np.array_equal(x0[b], x1[b]) and np.array_equal(v0[b], v1[b])
        vo = o.block_variances(b)
        assert np.abs(x1[b] - o.block_estimates(b)).max() < 1e-8 and np.abs(v1[b] - vo).max() / np.abs(vo).max() < 1e-8
    a1.GenerateStatistics()
    assert abs(a1.GetSigmaZero() - o.statistics()[0].sigma_zero) < 1e-9
    a1.close()
    o.close()


def test_batch_size_is_capped(built, tmp_path):
    """batch_blocks = 2: groups of at most two members, same bits"""
    adjust.write_synthetic_network(str(tmp_path), "c", 48, 40, 0, 6, seed=5)
    runs = []
    for cap in (0, 2, 3):
        a, st = _run(str(tmp_path), "c", multi_thread=True, batch_blocks=cap)
        assert st == 0
        runs.append(_results(a) + (a.batched_block_steps(),))
        a.close()
    assert runs[0][3] == 0 and runs[1][3] > 0 and runs[2][3] > 0
    for r in runs[1:]:
        assert r[2] == runs[0][2]
        for b in range(len(r[0])):
            assert np.array_equal(r[0][b], runs[0][0][b]) and np.array_equal(r[1][b], runs[0][1][b])


def test_a_singular_member_is_reported_with_its_block(built, tmp_path):
    """a member whose normals are not positive definite: the reference's message (dnamatrix_contiguous.cpp:983 through SolveTry,
    dnaadjust.cpp:6575-6582), with the member's own block number"""
    from tests import dnaformats as F
    adjust.write_synthetic_network(str(tmp_path), "s", 48, 40, 0, 6, seed=8)
    base = os.path.join(str(tmp_path), "s")
    # an inner station of block 4 loses every measurement, and the free stations' constraint weight underflows to zero: a zero pivot
    ISL, JSL, CML, nets = F.read_seg(base + ".seg")
    msr = F.read_bms(base + ".bms")
    target = int(ISL[3][len(ISL[3]) // 2])
    CML[3] = np.array([int(i) for i in CML[3] if int(msr[int(i)]["station1"]) != target and int(msr[int(i)]["station2"]) != target], dtype=np.uint32)
    F.write_seg(base + ".seg", ISL, JSL, CML, nets, msr)
    for cap in (0, 16):
        with pytest.raises(adjust.NetAdjustException) as e:
            _run(str(tmp_path), "s", multi_thread=True, batch_blocks=cap, free_std_dev=1e200)
        assert "singular" in str(e.value) and "block 4" in str(e.value), str(e.value)


def test_without_room_for_the_members_the_blocks_go_one_at_a_time(built, tmp_path):
    """the members' workspaces cannot be allocated (injected): every group falls back to the unbatched calls, same bits, no failure"""
    adjust.write_synthetic_network(str(tmp_path), "m", 48, 40, 0, 6, seed=13)
    a0, st0 = _run(str(tmp_path), "m", multi_thread=True, batch_blocks=0)
    ref = _results(a0)
    a0.close()
    built.dnagpu_debug_fail_batch_workspaces(1000)
    try:
        a1, st1 = _run(str(tmp_path), "m", multi_thread=True, batch_blocks=16)
        assert st1 == st0 == 0 and a1.batched_block_steps() == 0
        got = _results(a1)
        a1.close()
    finally:
        built.dnagpu_debug_fail_batch_workspaces(0)
    assert got[2] == ref[2]
    for b in range(len(ref[0])):
        assert np.array_equal(ref[0][b], got[0][b]) and np.array_equal(ref[1][b], got[1][b])
    # and with room again the same adjustment is batched
    a2, st2 = _run(str(tmp_path), "m", multi_thread=True, batch_blocks=16)
    assert st2 == 0 and a2.batched_block_steps() > 0
    a2.close()
