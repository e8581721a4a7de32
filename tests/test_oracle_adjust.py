"""CPU oracle: the restated phased chain against (a) the committed golden network and (b) the
reference's own claim that the phased result is rigorous (== the simultaneous solution)."""
import os

import numpy as np
import pytest

from dynadjust_amd.device import unpack_lower
from tests import dnaformats as F


def _run(orc, base, phased, **kw):
    net = orc.Network(base, phased)
    a = orc.Adjustment(net, phased, **kw)
    a.prepare()
    st = a.run()
    return net, a, st


def _phased_vs_simultaneous(orc, base, tol_x=5e-9, tol_v=1e-9):
    ns, s, st_s = _run(orc, base, False)
    npn, p, st_p = _run(orc, base, True)
    assert st_s == 0 and st_p == 0
    xs = s.block_estimates(0).reshape(-1, 3)
    Vs = unpack_lower(s.block_variances(0), 3 * ns.n_stations)
    for b in range(p.n_blocks):
        st = p.block_stations(b)
        xb = p.block_estimates(b).reshape(-1, 3)
        assert np.abs(xb - xs[st]).max() < tol_x
        idx = (3 * st[:, None] + np.arange(3)).ravel()
        Vb = unpack_lower(p.block_variances(b), 3 * len(st))
        assert np.abs(Vb - Vs[np.ix_(idx, idx)]).max() / np.abs(Vs).max() < tol_v
    s.close()
    p.close()


def test_golden_tiny_network(orc, golden_dir):
    exp = np.load(os.path.join(golden_dir, "tiny_net_expected.npz"))
    base = os.path.join(golden_dir, "tiny_net")
    for phased, tag in ((False, "simult"), (True, "phased")):
        net, a, st = _run(orc, base, phased)      # built-in LAPACK; the fixture was made with MKL
        assert st == int(exp[f"{tag}_status"])
        assert a.iterations() == int(exp[f"{tag}_iterations"])
        for b in range(a.n_blocks):
            assert np.array_equal(a.block_stations(b), exp[f"{tag}_stations_{b}"])
            assert np.abs(a.block_estimates(b) - exp[f"{tag}_estimates_{b}"]).max() < 1e-8
            v = exp[f"{tag}_variances_{b}"]
            assert np.abs(a.block_variances(b) - v).max() / np.abs(v).max() < 1e-9
        a.close()


@pytest.mark.parametrize("rows,cols,nbl,blocks", [(8, 6, 0, 2), (10, 7, 150, 3), (12, 9, 0, 6), (9, 9, 160, 9)])
def test_phased_is_rigorous(orc, built, tmp_path, rows, cols, nbl, blocks):
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "n", rows, cols, nbl, blocks, seed=rows * 100 + blocks)
    _phased_vs_simultaneous(orc, str(tmp_path / "n"))


def test_adjustment_recovers_the_truth(orc, built, tmp_path):
    """corner stations constrained at their true coordinates: the adjusted network must sit on the truth
    within the noise of the observations (3-6 mm per baseline component)"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "n", 10, 10, 0, 4, initial_sigma=1e-9)
    base = str(tmp_path / "n")
    truth = np.fromfile(base + ".truth").reshape(-1, 3)
    net, a, st = _run(orc, base, True)
    assert st == 0
    for b in range(a.n_blocks):
        xb = a.block_estimates(b).reshape(-1, 3)
        assert np.abs(xb - truth[a.block_stations(b)]).max() < 0.03
    a.close()


def test_multiple_networks_and_isolated_blocks(orc, built, tmp_path):
    """two contiguous networks + one isolated single-block network in one project (blockMeta
    first/last/isolated, dnaadjust.cpp:10449-10474): every network must equal its own simultaneous solution"""
    from dynadjust_amd import adjust
    specs = [("a", 8, 5, 3), ("b", 5, 5, 1), ("c", 6, 6, 2)]
    for nm, r, c, blk in specs:
        adjust.write_synthetic_network(str(tmp_path), nm, r, c, 0, blk, seed=ord(nm))
    F.merge_networks([str(tmp_path / s[0]) for s in specs], str(tmp_path / "all"))
    net, p, st = _run(orc, str(tmp_path / "all"), True)
    assert st == 0 and p.n_blocks == 6
    off = 0
    b0 = 0
    for nm, r, c, blk in specs:
        _, s, st_s = _run(orc, str(tmp_path / nm), False)
        xs = s.block_estimates(0).reshape(-1, 3)
        for b in range(b0, b0 + blk):
            stn = p.block_stations(b) - off
            assert np.abs(p.block_estimates(b).reshape(-1, 3) - xs[stn]).max() < 5e-9
        off += r * c
        b0 += blk
        s.close()
    p.close()


def test_iteration_limit_status(orc, golden_dir):
    base = os.path.join(golden_dir, "tiny_net")
    _, a, st = _run(orc, base, True, max_iterations=1)
    assert st == 1 and a.iterations() == 1          # ADJUST_MAX_ITERATIONS_EXCEEDED (dnaadjust.cpp:2526-2528)
    a.close()


def test_singular_variance_matrix_is_reported(orc, golden_dir):
    net = orc.Network(os.path.join(golden_dir, "tiny_net"), True)
    net.vcv6 = net.vcv6.copy()
    net.vcv6[0:6] = [1.0, 2.0, 1.0, 0.0, 0.0, 1.0]
    a = orc.Adjustment(net, True)
    with pytest.raises(RuntimeError) as e:
        a.prepare()
    assert "singular" in str(e.value)
    a.close()


def test_scale_normals_to_unity_gives_same_solution(orc, golden_dir):
    base = os.path.join(golden_dir, "tiny_net")
    _, a, _ = _run(orc, base, True)
    _, b, _ = _run(orc, base, True, scale_normals_to_unity=True)
    for k in range(a.n_blocks):
        assert np.abs(a.block_estimates(k) - b.block_estimates(k)).max() < 1e-8
    a.close()
    b.close()
