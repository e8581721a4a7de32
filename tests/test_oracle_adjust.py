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


@pytest.mark.parametrize("kw", [{"ragged": 0.6, "n_blocks": 5}, {"rows_lo": 1, "rows_hi": 4}, {"rows_lo": 2, "rows_hi": 2}])
def test_uneven_segmentations_are_rigorous(orc, built, tmp_path, kw):
    """strips of uneven height (dnasynth_spec.ragged / rows_lo / rows_hi -- what dnasegment makes of a real network: no two blocks alike,
    dnasegment.cpp:235-348): the segmentation invariants hold (every station inner in exactly one block, JSL(k) inside block k + 1,
    every measurement in one block) and the phased result equals the simultaneous one"""
    from dynadjust_amd import adjust
    info = adjust.write_synthetic_network(str(tmp_path), "u", 14, 7, 0, kw.get("n_blocks", 1), seed=77, **{k: v for k, v in kw.items() if k != "n_blocks"})
    net = orc.Network(str(tmp_path / "u"), True)
    assert net.n_blocks == info["blocks"] and (info["blocks"] == 5 if "ragged" in kw else info["blocks"] >= 4)
    sizes = [len(net.block_inner(b)) for b in range(net.n_blocks)] if hasattr(net, "block_inner") else None
    if "ragged" in kw and sizes:
        assert len(set(sizes)) > 1
    _phased_vs_simultaneous(orc, str(tmp_path / "u"))


def _fuzzed_project(tmp_path, seed):
    """one randomly cut network (persistent junctions, a last block without measurements of its own), a second one with another cut, and
    an isolated single-block network, merged into one project with three network ids (tests/segfuzz.py)"""
    from dynadjust_amd import adjust
    from tests import segfuzz
    rng = np.random.default_rng(1000 + seed)
    info = []
    for nm, rows, cols, mean in (("a", int(rng.integers(7, 12)), int(rng.integers(6, 11)), int(rng.integers(4, 14))),
                                 ("b", int(rng.integers(5, 8)), int(rng.integers(5, 8)), int(rng.integers(3, 9)))):
        adjust.write_synthetic_network(str(tmp_path), nm, rows, cols, 0, 1, seed=seed * 7 + ord(nm), initial_sigma=0.2)
        info.append(segfuzz.write_cut(str(tmp_path / nm), rng, mean_block=mean, noise=float(rng.uniform(0.1, 0.6)), lone_last=bool(seed % 2)))
    adjust.write_synthetic_network(str(tmp_path), "c", 4, 4, 0, 1, seed=seed + 3)
    F.merge_networks([str(tmp_path / nm) for nm in "abc"], str(tmp_path / "all"))
    return info


@pytest.mark.parametrize("seed", range(8))
def test_random_segmentations_are_rigorous(orc, built, tmp_path, seed):
    """dnasegment-like cuts that the strip generator cannot make (tests/segfuzz.py): junction stations that STAY junction over several
    blocks (dnasegment.cpp:529-531), junction sets of uneven size, blocks without a measurement of their own, two contiguous networks and
    an isolated block in one project (dnaadjust.cpp:10449-10474) -- every network of the phased result equals its own simultaneous solution"""
    info = _fuzzed_project(tmp_path, seed)
    assert info[0]["max_junction_life"] >= 2 and info[0]["blocks"] >= 4
    net, p, st = _run(orc, str(tmp_path / "all"), True)
    assert st == 0 and p.n_blocks == info[0]["blocks"] + info[1]["blocks"] + 1
    off = b0 = 0
    for nm, blk in (("a", info[0]["blocks"]), ("b", info[1]["blocks"]), ("c", 1)):
        ns, s, st_s = _run(orc, str(tmp_path / nm), False)
        xs = s.block_estimates(0).reshape(-1, 3)
        Vs = unpack_lower(s.block_variances(0), 3 * ns.n_stations)
        for b in range(b0, b0 + blk):
            stn = p.block_stations(b) - off
            assert np.abs(p.block_estimates(b).reshape(-1, 3) - xs[stn]).max() < 5e-9
            idx = (3 * stn[:, None] + np.arange(3)).ravel()
            Vb = unpack_lower(p.block_variances(b), 3 * len(stn))
            assert np.abs(Vb - Vs[np.ix_(idx, idx)]).max() / np.abs(Vs).max() < 1e-9
        off += ns.n_stations
        b0 += blk
        s.close()
    p.close()


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
    Wc = np.zeros((n, n))
    for s in range(net.n_stations):
        cst = net.constraints[3 * s:3 * s + 3]
        v = [(fixed_std_dev if c == ord("C") else free_std_dev) ** 2 for c in cst]
        if cst in (b"CCC", b"FFF"):
            Wc[3 * s:3 * s + 3, 3 * s:3 * s + 3] = np.eye(3) / v[0]
            continue
        # mixed codes (FormConstraintStationVarianceMatrix, dnaadjust.cpp:2041): per-axis variances in the local frame --
        # (latitude, longitude, up) for geographic records, (east, north, up) for projection records, x / y / z for cartesian
        # ones -- rotated to cartesian with the record's position, then inverted
        t = int(net._supplied_type[s])
        if t == 0:
            V = np.diag(v)
        else:
            e_, n_, u_ = (v[1], v[0], v[2]) if t in (1, 2) else (v[0], v[1], v[2])
            lat, lon = net._llh[s][0], net._llh[s][1]
            R = np.array([[-np.sin(lon), -np.sin(lat) * np.cos(lon), np.cos(lat) * np.cos(lon)],
                          [np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat) * np.sin(lon)],
                          [0.0, np.cos(lat), np.sin(lat)]])
            V = R @ np.diag([e_, n_, u_]) @ R.T
        Wc[3 * s:3 * s + 3, 3 * s:3 * s + 3] = np.linalg.inv(V)
    N = A.T @ W @ A + Wc
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


@pytest.mark.parametrize("rows,cols,blocks,xcl,ycl", [(10, 8, 3, 0, False), (10, 8, 3, 6, False), (9, 12, 4, 1000, True)])
def test_against_the_extended_precision_solution(orc, built, tmp_path, rows, cols, blocks, xcl, ycl):
    """The reference's fp64 solver cannot run here, so the 1e-8 m claim is argued through the EXACT answer: tests/exact.py solves the network
    in numpy longdouble (64-bit mantissa) from the dense design and weight matrices to a Cholesky of its own -- nothing shared with the oracle
    but the file reader.  The oracle's fp64 results, simultaneous and phased, are within two units in the last place of a 4e6 m coordinate
    (1.9e-9 m) of it and its variance matrix within 1e-12 relative: any other correct fp64 solution -- the reference's -- is then within
    4e-9 m of the oracle's."""
    from dynadjust_amd import adjust
    from tests import exact
    adjust.write_synthetic_network(str(tmp_path), "e", rows, cols, 0, blocks, seed=31 + rows, x_clusters=xcl, y_cluster=ycl)
    base = str(tmp_path / "e")
    net, a, st = _run(orc, base, False)
    x, V, its = exact.solve(net)
    assert st == 0 and a.iterations() == its
    assert float(np.abs(np.asarray(a.block_estimates(0), dtype=np.longdouble) - x).max()) < 1.9e-9
    Vo = unpack_lower(a.block_variances(0), 3 * net.n_stations)
    assert float(np.abs(np.asarray(Vo, dtype=np.longdouble) - V).max() / np.abs(V).max()) < 1e-12
    a.close()
    netp, p, st = _run(orc, base, True)
    assert st == 0
    for b in range(p.n_blocks):
        stn = p.block_stations(b)
        xb = np.asarray(p.block_estimates(b), dtype=np.longdouble).reshape(-1, 3)
        assert float(np.abs(xb - x.reshape(-1, 3)[stn]).max()) < 1.9e-9
        idx = (3 * stn[:, None] + np.arange(3)).ravel()
        Vb = np.asarray(unpack_lower(p.block_variances(b), 3 * len(stn)), dtype=np.longdouble)
        assert float(np.abs(Vb - V[np.ix_(idx, idx)]).max() / np.abs(V).max()) < 1e-11
    p.close()


def _local_sd(stn, V):
    """sqrt of the diagonal of R V R^T per station (e, n, up), V = full variance matrix in station order"""
    out = []
    for i, s in enumerate(stn):
        lat, lon = s[2], s[3]
        R = np.array([[-np.sin(lon), np.cos(lon), 0.0],
                      [-np.sin(lat) * np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat)],
                      [np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)]])
        out.append(np.sqrt(np.diag(R @ V[3 * i:3 * i + 3, 3 * i:3 * i + 3] @ R.T)))
    return np.array(out)


def check_against_reference_report(adj, names, stn, xyz, V, fields, stats):
    """the reference's own adjustment report of its sample network (gnss.simult.adj.expected: 4-decimal tables, from
    4-decimal observations, hence the tolerances)"""
    exp_xyz = np.array([adj["stn"][n]["xyz"] for n in names])
    assert np.abs(xyz.reshape(-1, 3) - exp_xyz).max() < 2.5e-4
    exp_sd = np.array([adj["stn"][n]["sd_enu"] for n in names])
    assert np.abs(_local_sd(stn, V) - exp_sd).max() < 1e-4
    col = lambda k: np.array([m[k] for m in adj["msr"]])
    assert np.abs(fields["measAdj"] - col("adjusted")).max() < 2.5e-4
    assert np.abs(fields["measCorr"] - col("correction")).max() < 2.5e-4
    assert np.abs(np.sqrt(fields["measPrec"]) - col("meas_sd")).max() < 1e-4
    assert np.abs(np.sqrt(fields["measAdjPrec"]) - col("adj_sd")).max() < 1e-4
    assert np.abs(np.sqrt(fields["residualPrec"]) - col("corr_sd")).max() < 1e-4
    assert np.abs(fields["NStat"] - col("nstat")).max() < 0.06
    assert stats["measurements"] == adj["measurements"] == 417 and stats["unknowns"] == adj["unknowns"] == 129
    assert stats["dof"] == adj["dof"] == 288
    assert abs(stats["chi2"] - adj["chi2"]) < 0.5            # 336.64 in the report
    assert abs(stats["sigma0"] - adj["sigma0"]) < 2e-3       # 1.169
    assert abs(stats["pelzer"] - 0.779) < 1e-3               # "Global (Pelzer) Reliability 0.779"
    assert stats["outliers"] == 10                           # "(10 potential outliers)"


def test_oracle_against_the_exact_solution_at_size(orc, built, golden_dir, tmp_path):
    """the oracle, phased (3 blocks of n = 1 500, junctions of 300 unknowns, 1e12 constraint weights), against the committed extended-precision
    solution of the same network of n = 3 600 unknowns (tests/golden/exact_3k.npz, tools/make_exact_golden.py; tests/test_gpu_exact.py holds
    the device to the same record): 1e-8 m, 1e-8 relative -- measured: a few units in the last place of a 4e6 m coordinate"""
    import json
    from dynadjust_amd import adjust
    from tests.test_gpu_exact import compare_with_exact
    g = np.load(os.path.join(golden_dir, "exact_3k.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    info = adjust.write_synthetic_network(str(tmp_path), "e", meta["rows"], meta["cols"], 0, meta["blocks"], seed=meta["seed"])
    assert info["stations"] == meta["stations"]
    fast = orc.scipy_openblas_path()
    if fast:
        orc.use_lapack(fast)
    try:
        net, p, st = _run(orc, str(tmp_path / "e"), True, threads=8)
    finally:
        orc.use_mkl(False)
    assert st == 0 and p.iterations() == meta["iterations"]
    rec = compare_with_exact(g, meta, p.n_blocks, p.block_stations, p.block_estimates, p.block_variances)
    assert rec["max_abs_dx_m"] < 5e-9 and rec["max_rel_dvar_diagonal"] < 1e-9 and rec["max_rel_dvar_sampled_columns"] < 1e-9, rec
    assert rec["max_rel_dfrobenius"] < 1e-9 and rec["max_rel_dquadratic_forms"] < 1e-8, rec
    p.close()


@pytest.mark.parametrize("importer", ["product", "test"])
def test_reference_sample_gnss_network(orc, built, golden_dir, tmp_path, importer):
    """the oracle against the reference's published result for sampleData/gnss-network (129 G, 1 X cluster of 4,
    1 Y cluster of 6, variance scalars): this pins the restated adjustment end to end.  "product": the sample's .stn / .msr
    through the product's importer (DNA text reader + frame alignment, host/dnaimport_lite.cpp; no GPU involved);
    "test": the test-only reader with the report's "Measured" column as observations."""
    from tests import dnatext as T
    base = str(tmp_path / "gnss")
    stn, cl, adj = (T.build_gnss_sample_with_the_product_importer if importer == "product" else T.build_gnss_sample)(golden_dir, base)
    net, a, st = _run(orc, base, False)
    assert st == 0 and a.iterations() == 2                    # "ITERATION 2 ... SOLUTION Converged"
    s, f = a.statistics()
    V = unpack_lower(a.block_variances(0), 3 * len(stn))
    stats = {"measurements": s.measurement_params, "unknowns": s.unknown_params, "dof": s.dof, "chi2": s.chi_squared,
             "sigma0": s.sigma_zero, "pelzer": s.global_pelzer, "outliers": s.potential_outliers}
    check_against_reference_report(adj, [x[0] for x in stn], stn, a.block_estimates(0), V, f, stats)
    a.close()


def test_statistics_phased_equals_simultaneous(orc, built, tmp_path):
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", 9, 7, 0, 3, seed=5, x_clusters=10, y_cluster=True)
    base = str(tmp_path / "s")
    ns, s, st_s = _run(orc, base, False)
    npn, p, st_p = _run(orc, base, True)
    ss, fs = s.statistics()
    sp, fp = p.statistics()
    assert abs(ss.chi_squared - sp.chi_squared) < 1e-6 * ss.chi_squared
    assert (ss.dof, ss.potential_outliers) == (sp.dof, sp.potential_outliers)
    assert abs(ss.global_pelzer - sp.global_pelzer) < 1e-6
    for k in ("measAdj", "measCorr"):
        assert np.abs(fs[k] - fp[k]).max() < 1e-8
    for k in ("measAdjPrec", "residualPrec"):
        assert np.abs(fs[k] - fp[k]).max() < 1e-6 * np.abs(fs[k]).max()   # ~1e-9 relative agreement of the block variances
    s.close()
    p.close()


def test_quantiles_against_scipy(built):
    """the quantile functions behind the global test (boost::math in the reference) against scipy"""
    from scipy.stats import chi2, norm
    from dynadjust_amd import _lib
    lib = _lib.load()
    for p in (1e-9, 1e-4, 0.01, 0.025, 0.3, 0.5, 0.8, 0.975, 0.995, 1 - 1e-7):
        # (p itself carries half an ulp: 1.1e-16 / pdf in x)
        assert abs(lib.dnastat_normal_quantile(p) - norm.ppf(p)) < 1e-12 + 4e-16 / norm.pdf(norm.ppf(p))
        for dof in (1, 2, 3, 7, 30, 288, 5000, 240000):
            q, r = lib.dnastat_chi_squared_quantile(dof, p), chi2.ppf(p, dof)
            assert abs(q - r) < 1e-9 * max(r, 1e-3) + 4e-16 / max(chi2.pdf(r, dof), 1e-300), (dof, p, q, r)


def test_vcv_frames_and_scalars(orc):
    """PropagateVariances_GeoCart_Cluster / ScaleGPSVCV_Cluster (dnatemplatematrixfuncs.hpp:355-443) restated in the oracle,
    against independent formulas: the Jacobian of GeoToCart by central differences, and the fact that scaling in the
    geographic frame is a scaling along the local north / east / up axes (the metric factors of J commute with the
    diagonal scalars): V' = (R S R^T) V (R S R^T)^T."""
    rng = np.random.default_rng(3)
    k = 3
    llh = np.array([[-0.64, 2.55, 210.0], [-0.63, 2.56, 890.0], [-0.65, 2.54, 15.0]])
    A = rng.standard_normal((3 * k, 3 * k + 2))
    V = A @ A.T * 1e-5
    J = np.zeros((3 * k, 3 * k))
    R = np.zeros((3 * k, 3 * k))
    for a in range(k):
        lat, lon, h = llh[a]
        for c, d in enumerate((1e-7, 1e-7, 1e-2)):
            p, m = llh[a].copy(), llh[a].copy()
            p[c] += d
            m[c] -= d
            J[3 * a:3 * a + 3, 3 * a + c] = (np.array(orc.geo_to_cart(*p)) - np.array(orc.geo_to_cart(*m))) / (2 * d)
        # columns: north, east, up unit vectors
        R[3 * a:3 * a + 3, 3 * a:3 * a + 3] = np.array([
            [-np.sin(lat) * np.cos(lon), -np.sin(lon), np.cos(lat) * np.cos(lon)],
            [-np.sin(lat) * np.sin(lon), np.cos(lon), np.cos(lat) * np.sin(lon)],
            [np.cos(lat), 0.0, np.sin(lat)]])
    Vg = A @ A.T * np.outer(np.tile([1e-9, 1e-9, 1e-2], k), np.tile([1e-9, 1e-9, 1e-2], k))   # a geographic-frame matrix
    got = orc.propagate_geo_cart(Vg, llh, True)
    ref = J @ Vg @ J.T
    assert np.abs(got - ref).max() < 1e-6 * np.abs(ref).max()
    back = orc.propagate_geo_cart(got, llh, False)
    assert np.abs(back - Vg).max() / np.abs(Vg).max() < 1e-9
    p, l, hh = 2.0, 3.0, 0.5
    S = np.diag(np.tile(np.sqrt([p, l, hh]), k))
    M = R @ S @ R.T
    ref = M @ V @ M.T
    got = orc.scale_gps_vcv(V, llh, p, l, hh, False)
    assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()
    assert np.abs(got - got.T).max() < 1e-12 * np.abs(got).max()
    # already geographic input: only the second half of the chain
    got = orc.scale_gps_vcv(Vg, llh, p, l, hh, True)
    ref = J @ S @ Vg @ S @ J.T
    assert np.abs(got - ref).max() < 1e-6 * np.abs(ref).max()


@pytest.mark.parametrize("blocks,ycl", [(1, True), (3, True), (2, False)])
def test_gnss_scalars_and_llh_point_clusters(orc, built, tmp_path, blocks, ycl):
    """variance scalars (v, phi, lambda, h) on G / X / Y and Y clusters supplied as latitude / longitude / height"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", 8, 6, 0, blocks, seed=21, x_clusters=9, y_cluster=ycl, y_llh=ycl, scalars=True)
    base = str(tmp_path / "s")
    bms = F.read_bms(base + ".bms")
    if ycl:
        assert {bytes(c).rstrip(b"\x00") for c in bms["coordType"][bms["measType"] == b"Y"]} == {b"LLh", b"LLH"}
    assert np.any(bms["scale1"] != 1.0) and np.any(bms["scale4"] != 1.0)
    net, a, st = _run(orc, base, False)
    assert st == 0
    x, V = _dense_solution(net)
    assert np.abs(a.block_estimates(0) - x).max() < 2e-8
    if ycl:
        truth = np.fromfile(base + ".truth", dtype=np.float64)
        assert np.abs(a.block_estimates(0) - truth).max() < 0.08      # the LLH -> XYZ conversion of the datum points worked
    # the scalars act: without them the solution differs
    plain = bms.copy()
    for kf in ("scale1", "scale2", "scale3", "scale4"):
        plain[kf] = 1.0
    F.write_bms(str(tmp_path / "p.bms"), plain)
    for ext in ("bst", "asl", "seg"):
        import shutil
        shutil.copy(base + "." + ext, str(tmp_path / ("p." + ext)))
    netp, ap, stp = _run(orc, str(tmp_path / "p"), False)
    assert np.abs(ap.block_estimates(0) - a.block_estimates(0)).max() > 1e-6
    a.close()
    ap.close()
    if blocks > 1:
        _phased_vs_simultaneous(orc, base)


@pytest.mark.parametrize("blocks", [1, 3])
def test_mixed_station_constraints(orc, built, tmp_path, blocks):
    """CCF / CFF / FFC / CFC station constraints on geographic, projection and cartesian station records
    (FormConstraintStationVarianceMatrix, dnaadjust.cpp:2041-2137): the oracle against the dense numpy solution"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "m", 7, 6, 0, blocks, seed=31, x_clusters=6)
    base = str(tmp_path / "m")
    bst = F.read_bst(base + ".bst").copy()
    for s, (code, typ) in {3: (b"CCF", 2), 8: (b"FFC", 1), 14: (b"CFF", 3), 20: (b"CFC", 3), 27: (b"FCC", 0), 33: (b"FFC", 0)}.items():
        bst["stationConst"][s] = code
        bst["suppliedStationType"][s] = typ
    F.write_bst(base + ".bst", bst)
    net, a, st = _run(orc, base, blocks > 1)
    assert st == 0
    x, V = _dense_solution(net)
    for k in range(a.n_blocks):
        stn = a.block_stations(k)
        idx = (3 * stn[:, None] + np.arange(3)).ravel()
        assert np.abs(a.block_estimates(k) - x[idx]).max() < 2e-8
        Vb = unpack_lower(a.block_variances(k), 3 * len(stn))
        assert np.abs(Vb - V[np.ix_(idx, idx)]).max() < 1e-7 * np.abs(V).max()
    # the constrained components do not move, the free ones do
    x0 = net.xyz0.reshape(-1, 3)
    moved = np.abs(x.reshape(-1, 3) - x0)
    assert moved[27, 1:].max() < 1e-6 and moved[27, 0] > 1e-4          # FCC on a cartesian record: Y and Z held
    a.close()
