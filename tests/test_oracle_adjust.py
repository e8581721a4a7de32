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


def _dense_solution(net, fixed_std_dev=1e-6, free_std_dev=10.0):
    """independent dense numpy solution of the (linear) GNSS network, iterated the way AdjustSimultaneous does
    (the station constraints weight the normals only, so every iteration pulls towards the previous estimates):
    x <- x + (A'WA + Wc)^-1 A'W (obs - A x) until the largest correction is below the threshold"""
    n = 3 * net.n_stations
    m = 3 * net.n_baselines
    A = np.zeros((m, n))
    for i in range(net.n_baselines):
        for c in range(3):
            if net.stn1[i] != 0xffffffff:
                A[3 * i + c, 3 * int(net.stn1[i]) + c] = -1.0
            A[3 * i + c, 3 * int(net.stn2[i]) + c] = 1.0
    W = np.zeros((m, m))
    voff = 0
    for c in range(net.n_clusters):
        i0, i1 = int(net.cluster_off[c]), int(net.cluster_off[c + 1])
        nc = 3 * (i1 - i0)
        V = net.cluster_vcv[voff:voff + nc * nc].reshape(nc, nc, order="F")
        voff += nc * nc
        W[3 * i0:3 * i1, 3 * i0:3 * i1] = np.linalg.inv(V)
    Wc = np.zeros(n)
    for s in range(net.n_stations):
        cst = net.constraints[3 * s:3 * s + 3]
        assert cst in (b"CCC", b"FFF")
        Wc[3 * s:3 * s + 3] = 1.0 / (fixed_std_dev if cst == b"CCC" else free_std_dev) ** 2
    N = A.T @ W @ A + np.diag(Wc)
    x = net.xyz0.copy()
    for _ in range(10):
        dx = np.linalg.solve(N, A.T @ W @ (net.obs - A @ x))
        x += dx
        if np.abs(dx).max() < float(np.float32(0.0005)):
            break
    return x, np.linalg.inv(N)


@pytest.mark.parametrize("rows,cols,blocks,xcl,ycl", [(6, 6, 1, 8, False), (7, 6, 3, 12, True), (8, 5, 4, 1000, True)])
def test_gnss_clusters(orc, built, tmp_path, rows, cols, blocks, xcl, ycl):
    """'X' baseline clusters and 'Y' point clusters (full 3k x 3k variance matrices, dnaadjust.cpp:4312/4494):
    the oracle against a dense numpy solution, and phased against simultaneous."""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "c", rows, cols, 0, blocks, seed=77 + rows, x_clusters=xcl, y_cluster=ycl)
    base = str(tmp_path / "c")
    bms = F.read_bms(base + ".bms")
    types = set(bytes(t) for t in bms["measType"])
    assert b"X" in types and (b"Y" in types) == ycl
    net, a, st = _run(orc, base, False)
    assert st == 0 and net.n_clusters > 0 and int(np.diff(net.cluster_off).max()) >= 2
    x, V = _dense_solution(net)
    assert np.abs(a.block_estimates(0) - x).max() < 2e-8
    Vo = unpack_lower(a.block_variances(0), 3 * net.n_stations)
    assert np.abs(Vo - V).max() / np.abs(V).max() < 1e-7
    if ycl:      # the datum comes from the point clusters (millimetre noise), not from the perturbed initial coordinates
        truth = np.fromfile(base + ".truth", dtype=np.float64)
        assert np.abs(a.block_estimates(0) - truth).max() < 0.05
    a.close()
    if blocks > 1:
        _phased_vs_simultaneous(orc, base)
