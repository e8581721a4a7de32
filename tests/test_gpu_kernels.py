"""Block-level kernels through the C-ABI against the CPU oracle on the same seeded network:
weights, meas-minus-computed, normal-equation formation (bit-exact), rhs and the junction carry."""
import os

import numpy as np
import pytest

from dynadjust_amd.device import pack_lower, unpack_lower

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(built, orc, tmp_path_factory):
    from dynadjust_amd import adjust
    d = tmp_path_factory.mktemp("k")
    adjust.write_synthetic_network(str(d), "n", 9, 8, 170, 3, seed=11)
    base = str(d / "n")
    net = orc.Network(base, True)
    a = orc.Adjustment(net, True)
    a.prepare()
    return net, a


def _upload_block(ctx, net, a, b, blk_id=0):
    st = a.block_stations(b)
    loc = {int(s): i for i, s in enumerate(st)}
    cml = net.cml[net.cml_off[b]:net.cml_off[b + 1]]
    s1 = np.array([loc[int(net.stn1[i])] for i in cml], dtype=np.uint32)
    s2 = np.array([loc[int(net.stn2[i])] for i in cml], dtype=np.uint32)
    obs = np.concatenate([net.obs[3 * i:3 * i + 3] for i in cml])
    vcv = np.concatenate([net.vcv6[6 * i:6 * i + 6] for i in cml])
    xyz = np.concatenate([net.xyz0[3 * s:3 * s + 3] for s in st])
    ctx.block_create(blk_id, len(st), len(cml))
    ctx.block_set_stations(blk_id, xyz)
    ctx.block_set_baselines(blk_id, s1, s2, obs, vcv)
    return st, cml, s1, s2


def test_weights_and_b_are_bit_exact(gpu_ctx, case):
    net, a = case
    st, cml, s1, s2 = _upload_block(gpu_ctx, net, a, 1)
    w = gpu_ctx.block_get_weights(0, len(cml)).reshape(-1, 6)
    wo = a.weights().reshape(-1, 6)[cml]
    assert np.array_equal(w, wo), np.abs(w - wo).max()
    gpu_ctx.block_compute_b(0)
    b = gpu_ctx.block_get_b(0, len(cml))
    assert np.array_equal(b, a.block_b(1)[:3 * len(cml)])
    gpu_ctx.block_destroy(0)


def test_normals_are_bit_exact(gpu_ctx, case, orc):
    """N = sum A^T W A in CML order (dnaadjust.cpp:1664-1684) + forward constraints (:1884)"""
    net, a = case
    for b in range(3):
        st, cml, s1, s2 = _upload_block(gpu_ctx, net, a, b)
        n = 3 * len(st)
        m = gpu_ctx.matrix(n)
        gpu_ctx.form_normals(0, m, len(st))
        # forward constraints: stations appearing for the first time in the forward direction
        seen = set()
        for bb in range(b):
            seen |= set(int(s) for s in a.block_stations(bb))
        first = [i for i, s in enumerate(st) if int(s) not in seen]
        w9 = []
        for i in first:
            c = net.constraints[3 * int(st[i]):3 * int(st[i]) + 3]
            w = 1.0 / (1e-6 ** 2) if c == b"CCC" else 1.0 / (10.0 ** 2)
            w9 += [w, 0, 0, 0, w, 0, 0, 0, w]
        gpu_ctx.add_diag3x3(m, np.array(first, dtype=np.uint32), np.array(w9))
        got = m.download_packed()
        assert np.array_equal(got, a.block_normals(b)), np.abs(got - a.block_normals(b)).max()
        m.close()
        gpu_ctx.block_destroy(0)


def test_solve_and_junction_carry(gpu_ctx, case, orc):
    """one forward step: solve block 0, gather/invert its junction block, scatter into block 1 and
    form block 1's right-hand side -- every intermediate against numpy on the oracle's inputs"""
    net, a = case
    st0, cml0, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    st1, cml1, _, _ = _upload_block(gpu_ctx, net, a, 1, blk_id=1)
    n0, n1 = 3 * len(st0), 3 * len(st1)
    N0 = unpack_lower(a.block_normals(0), n0)
    m0 = gpu_ctx.matrix(n0)
    m0.upload_packed(a.block_normals(0), n0)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    rhs = gpu_ctx.block_get_rhs(0, len(st0))
    # numpy rhs = A^T W b
    W = a.weights().reshape(-1, 6)
    b0 = a.block_b(0)
    loc0 = {int(s): i for i, s in enumerate(st0)}
    ref = np.zeros(n0)
    for k, i in enumerate(cml0):
        w = W[i]
        Wm = np.array([[w[0], w[1], w[3]], [w[1], w[2], w[4]], [w[3], w[4], w[5]]])
        wb = Wm @ b0[3 * k:3 * k + 3]
        ref[3 * loc0[int(net.stn1[i])]:3 * loc0[int(net.stn1[i])] + 3] -= wb
        ref[3 * loc0[int(net.stn2[i])]:3 * loc0[int(net.stn2[i])] + 3] += wb
    assert np.abs(rhs - ref).max() <= 1e-12 * np.abs(ref).max()
    m0.invert()
    Ninv = np.linalg.inv(N0)
    gpu_ctx.solve_corrections(0, m0)
    corr = gpu_ctx.block_get_corrections(0, len(st0))
    assert np.abs(corr - Ninv @ ref).max() < 1e-9
    mv, row = gpu_ctx.update_estimates(0)
    assert row == int(np.argmax(np.abs(corr))) and mv == corr[row]      # matrix_2d::compute_maximum_value
    est0 = gpu_ctx.block_get_stations(0, 1, len(st0))
    # junction stations of block 0 (JSL(0)) in both blocks
    jsl = net.jsl[net.jsl_off[0]:net.jsl_off[1]]
    idx0 = np.array([loc0[int(s)] for s in jsl], dtype=np.uint32)
    loc1 = {int(s): i for i, s in enumerate(st1)}
    idx1 = np.array([loc1[int(s)] for s in jsl], dtype=np.uint32)
    jm = gpu_ctx.matrix(3 * len(jsl))
    gpu_ctx.junction_gather(0, m0, idx0, jm)
    rows = (3 * idx0[:, None] + np.arange(3)).ravel()
    J = unpack_lower(jm.download_packed(), 3 * len(jsl))
    assert np.abs(J - Ninv[np.ix_(rows, rows)]).max() < 1e-9 * np.abs(Ninv).max()
    assert np.array_equal(gpu_ctx.junction_get_estimates(jm), est0[rows])
    jm.invert()
    WJ = unpack_lower(jm.download_packed(), 3 * len(jsl))
    m1 = gpu_ctx.matrix(n1)
    m1.upload_packed(a.block_normals(1), n1)
    gpu_ctx.junction_scatter(m1, idx1, jm)
    N1 = unpack_lower(m1.download_packed(), n1)
    rows1 = (3 * idx1[:, None] + np.arange(3)).ravel()
    exp = unpack_lower(a.block_normals(1), n1)
    exp[np.ix_(rows1, rows1)] += WJ
    assert np.abs(N1 - exp).max() <= 1e-12 * np.abs(exp).max()
    gpu_ctx.block_compute_b(1)
    gpu_ctx.form_rhs(1)
    r_before = gpu_ctx.block_get_rhs(1, len(st1))
    gpu_ctx.junction_rhs(1, idx1, jm)
    r_after = gpu_ctx.block_get_rhs(1, len(st1))
    x1 = gpu_ctx.block_get_stations(1, 1, len(st1))
    bj = est0[rows] - x1[rows1]
    add = np.zeros(n1)
    add[rows1] = WJ @ bj
    assert np.abs((r_after - r_before) - add).max() <= 1e-9 * max(1.0, np.abs(add).max())
    for m in (m0, m1, jm):
        m.close()
    gpu_ctx.block_destroy(0)
    gpu_ctx.block_destroy(1)


def test_bad_arguments_are_rejected(gpu_ctx):
    from dynadjust_amd._lib import DnaGpuError
    gpu_ctx.block_create(0, 4, 2)
    with pytest.raises(DnaGpuError):   # station index out of range
        gpu_ctx.block_set_baselines(0, [0, 9], [1, 2], np.zeros(6), np.tile([1.0, 0, 1, 0, 0, 1], 2))
    with pytest.raises(DnaGpuError) as e:   # singular variance matrix -> same message as the reference's inverse failure
        gpu_ctx.block_set_baselines(0, [0, 1], [1, 2], np.zeros(6), np.array([1.0, 2, 1, 0, 0, 1, 1, 0, 1, 0, 0, 1]))
    assert "singular" in str(e.value)
    gpu_ctx.block_destroy(0)
    with pytest.raises(DnaGpuError):
        gpu_ctx.block_compute_b(77)


@pytest.fixture
def carry_form(request, built, gpu_ctx):
    """dnagpu_schur_carry in its estimates form (0) or its information form (1, the default) -- a setting of the context"""
    old = built.dnagpu_ctx_set_info_carry(gpu_ctx.h, request.param)
    assert built.dnagpu_info_carry(gpu_ctx.h) == request.param
    yield request.param
    built.dnagpu_ctx_set_info_carry(gpu_ctx.h, old)


@pytest.mark.parametrize("carry_form", [0, 1], indirect=True)
@pytest.mark.parametrize("rows,cols,strips,pick", [(9, 8, 3, "jsl"), (40, 30, 2, "jsl"), (40, 30, 2, "scattered"), (12, 11, 2, "all_but_one"),
                                                   (43, 43, 2, "one")])
def test_schur_carry_equals_solve_gather_invert(gpu_ctx, built, orc, tmp_path, rows, cols, strips, pick, carry_form):
    """dnagpu_schur_carry (partial elimination of the inner unknowns) against the reference's sequence Solve ->
    gather the junction block of N^-1 -> invert it (CarryStnEstimatesandVariancesForward, dnaadjust.cpp:998-1128), on the
    device and in numpy.  Sizes straddle the 128-tile boundaries (n = 3 * stations of the first strip + junction row).
    Both forms of the result: the estimates form carries estimates + corrections, the information form the estimates the block
    was formed at and the reduced right-hand side -- what dnagpu_junction_rhs then adds is the same."""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, strips, seed=rows + cols)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, cml0, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    n0 = 3 * len(st0)
    loc0 = {int(s): i for i, s in enumerate(st0)}
    if pick == "jsl":
        stn = [loc0[int(s)] for s in net.jsl[net.jsl_off[0]:net.jsl_off[1]]]
    elif pick == "scattered":
        stn = list(range(1, len(st0), 7))[::-1]                   # any order, interleaved with the inner stations
    elif pick == "all_but_one":
        stn = list(range(1, len(st0)))
    else:
        stn = [len(st0) // 2]
    idx = np.array(stn, dtype=np.uint32)
    rws = (3 * idx[:, None] + np.arange(3)).ravel()
    N0 = unpack_lower(a.block_normals(0), n0)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    rhs = gpu_ctx.block_get_rhs(0, len(st0))
    x0 = gpu_ctx.block_get_stations(0, 1, len(st0))
    # the reference's sequence on the device
    m = gpu_ctx.matrix(n0)
    m.upload_packed(a.block_normals(0), n0)
    m.invert()
    gpu_ctx.solve_corrections(0, m)
    corr = gpu_ctx.block_get_corrections(0, len(st0))
    jm = gpu_ctx.matrix(3 * len(idx))
    gpu_ctx.junction_gather(0, m, idx, jm)
    jm.invert()
    W_ref = unpack_lower(jm.download_packed(), 3 * len(idx))
    # elimination
    m.upload_packed(a.block_normals(0), n0)
    js = gpu_ctx.matrix(3 * len(idx))
    gpu_ctx.schur_carry(0, m, idx, js)
    W = unpack_lower(js.download_packed(), 3 * len(idx))
    est = gpu_ctx.junction_get_estimates(js)
    scale = np.abs(W_ref).max()
    assert np.abs(W - W_ref).max() < 1e-9 * scale, np.abs(W - W_ref).max() / scale
    if carry_form == 0:
        assert np.abs(est - (x0[rws] + corr[rws])).max() < 1e-9
    else:
        assert np.array_equal(est, x0[rws])
    assert np.array_equal(gpu_ctx.block_get_stations(0, 1, len(st0)), x0)          # the block's estimates are untouched
    # numpy: Schur complement and the reduced system
    inner = np.setdiff1d(np.arange(n0), rws)
    S = N0[np.ix_(rws, rws)] - N0[np.ix_(rws, inner)] @ np.linalg.solve(N0[np.ix_(inner, inner)], N0[np.ix_(inner, rws)]) if len(inner) else N0[np.ix_(rws, rws)]
    assert np.abs(W - S).max() < 1e-9 * scale
    d = np.linalg.solve(N0, rhs)
    if carry_form == 0:
        assert np.abs(est - (x0[rws] + d[rws])).max() < 1e-9
    # what the receiving block adds to its right-hand side (here: the block itself, same estimates): W * corrections in either form
    before = gpu_ctx.block_get_rhs(0, len(st0))
    gpu_ctx.junction_rhs(0, idx, js)
    added = (gpu_ctx.block_get_rhs(0, len(st0)) - before)[rws]
    want = S @ d[rws]
    # (estimates form: the corrections come back as (x + dx) - x, an ulp of a 6 000 km coordinate each, times the weights)
    tol = 1e-9 * max(1.0, np.abs(want).max()) if carry_form else 2e-9 * np.abs(S).sum(axis=1).max()
    assert np.abs(added - want).max() <= tol, (np.abs(added - want).max(), np.abs(want).max(), tol)
    others = np.setdiff1d(np.arange(n0), rws)
    assert not np.any((gpu_ctx.block_get_rhs(0, len(st0)) - before)[others])
    gpu_ctx.form_rhs(0)
    # a second call with another station list (the reverse direction's) and back again: both orders stay cached
    idx2 = np.array(sorted(set(range(len(st0))) - set(stn))[:max(1, len(st0) // 5)], dtype=np.uint32)
    j2 = gpu_ctx.matrix(3 * len(idx2))
    m.upload_packed(a.block_normals(0), n0)
    gpu_ctx.schur_carry(0, m, idx2, j2)
    r2 = (3 * idx2[:, None] + np.arange(3)).ravel()
    assert np.abs(gpu_ctx.junction_get_estimates(j2) - (x0[r2] + (0 if carry_form else 1) * d[r2])).max() < 1e-9
    m.upload_packed(a.block_normals(0), n0)
    gpu_ctx.schur_carry(0, m, idx, js)
    assert np.array_equal(unpack_lower(js.download_packed(), 3 * len(idx)), W)      # deterministic
    for q in (m, jm, js, j2):
        q.close()
    gpu_ctx.block_destroy(0)


def test_schur_carry_reports_a_singular_block(gpu_ctx, built, orc, tmp_path):
    """an unknown without any weight (zero row / column) among the eliminated ones -> dpotrf-style failure, same text as
    dnagpu_invert (MatrixInversionFailure, dnamatrix_contiguous.cpp:983)"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", 9, 8, 0, 2, seed=3)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, cml0, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    n0 = 3 * len(st0)
    N = unpack_lower(a.block_normals(0), n0)
    N[7, :] = 0.0
    N[:, 7] = 0.0
    m = gpu_ctx.matrix(n0)
    m.upload_packed(pack_lower(N), n0)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    jm = gpu_ctx.matrix(3)
    with pytest.raises(Exception) as e:
        gpu_ctx.schur_carry(0, m, np.array([0], dtype=np.uint32), jm)
    assert "singular" in str(e.value)
    m.close()
    jm.close()
    gpu_ctx.block_destroy(0)


@pytest.mark.parametrize("rows,cols,pick", [(9, 8, "jsl"), (40, 30, "jsl"), (40, 30, "scattered"), (43, 43, "one")])
def test_kept_factor_in_the_storage_of_the_result(gpu_ctx, built, orc, tmp_path, rows, cols, pick):
    """dnagpu_partial_create_in: the factor's inverse waits in the matrix that later receives the completed inverse (the block's rigorous
    variance matrix is dead between the start of an iteration and its rigorous solve) -- same bits as with storage of its own, the
    matrix reports itself empty meanwhile, and the lender may be the result"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, 2, seed=rows)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, _, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    ns = len(st0)
    n0 = 3 * ns
    loc0 = {int(s): i for i, s in enumerate(st0)}
    stn = {"jsl": [loc0[int(s)] for s in net.jsl[net.jsl_off[0]:net.jsl_off[1]]], "scattered": list(range(1, ns, 7))[::-1], "one": [ns // 2]}[pick]
    idx = np.array(stn, dtype=np.uint32)
    nk = 3 * len(idx)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    m = gpu_ctx.matrix(n0)
    red, kk = gpu_ctx.matrix(nk), gpu_ctx.matrix(nk)
    results = []
    for own in (True, False):
        store = gpu_ctx.matrix(n0 + 256)
        inv = gpu_ctx.matrix(n0) if own else store
        pf = gpu_ctx.partial_create(n0, nk) if own else gpu_ctx.partial_create_in(n0, nk, store)
        m.upload_packed(a.block_normals(0), n0)
        gpu_ctx.block_reduce(0, m, idx, red, keep=pf)
        S = unpack_lower(red.download_packed(), nk)
        kk.upload_packed(pack_lower(S + np.eye(nk) * np.abs(np.diag(S)).mean() * 0.05), nk)
        gpu_ctx.partial_complete(pf, kk, inv, n0)
        results.append(inv.download_packed().copy())
        if not own:
            with pytest.raises(Exception):                      # nothing is left of the factor: the storage went back to its matrix
                gpu_ctx.partial_reduce_rhs(0, pf, red)
        gpu_ctx.partial_destroy(pf)
        store.close()
        if own:
            inv.close()
    assert np.array_equal(results[0], results[1])
    tiny = gpu_ctx.matrix(max(3, n0 // 2))                       # a lender that cannot hold the factor is refused
    with pytest.raises(Exception):
        gpu_ctx.partial_create_in(n0, nk, tiny)
    for q in (m, red, kk, tiny):
        q.close()
    gpu_ctx.block_destroy(0)


@pytest.mark.parametrize("rows,cols,pick", [(9, 8, "jsl"), (40, 30, "jsl"), (40, 30, "scattered"), (43, 43, "one"), (60, 50, "jsl")])
def test_light_kept_factor(gpu_ctx, built, orc, tmp_path, rows, cols, pick):
    """dnagpu_partial_create_spine: the elimination keeps its block factor only (no inverse of the eliminated part); the kept block is
    factored (complete_factor), right-hand sides are solved by blocked substitution (partial_solve), and the inverse of the whole block is
    formed at the end (partial_finish) -- against numpy, and against the full-form partial"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, 2, seed=rows)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, _, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    ns = len(st0)
    n0 = 3 * ns
    loc0 = {int(s): i for i, s in enumerate(st0)}
    stn = {"jsl": [loc0[int(s)] for s in net.jsl[net.jsl_off[0]:net.jsl_off[1]]], "scattered": list(range(1, ns, 7))[::-1], "one": [ns // 2]}[pick]
    idx = np.array(stn, dtype=np.uint32)
    nk = 3 * len(idx)
    rws = (3 * idx[:, None] + np.arange(3)).ravel()
    N0 = unpack_lower(a.block_normals(0), n0)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    m = gpu_ctx.matrix(n0)
    red0, red, kk = gpu_ctx.matrix(nk), gpu_ctx.matrix(nk), gpu_ctx.matrix(nk)
    m.upload_packed(a.block_normals(0), n0)
    gpu_ctx.block_reduce(0, m, idx, red0)                                   # the plain elimination: same complement, same reduced rhs
    eye = gpu_ctx.matrix(n0)                                                # the block's right-hand side, read back through I * rhs
    eye.upload_packed(pack_lower(np.eye(n0)), n0)
    gpu_ctx.solve_corrections(0, eye)
    gpu_ctx.sync()
    rhs = gpu_ctx.block_get_corrections(0, ns).ravel().copy()
    eye.close()
    store = gpu_ctx.matrix(n0 + 256)
    pf = gpu_ctx.partial_create_spine(n0, nk, store)
    for rep in range(2):                                                    # (twice: the second elimination overwrites the first one's factor)
        m.upload_packed(a.block_normals(0), n0)
        gpu_ctx.block_reduce(0, m, idx, red, keep=pf)
        S = unpack_lower(red.download_packed(), nk)
        S0 = unpack_lower(red0.download_packed(), nk)
        assert np.abs(S - S0).max() < 1e-10 * np.abs(S0).max()
        assert np.abs(gpu_ctx.junction_get_estimates(red) - gpu_ctx.junction_get_estimates(red0)).max() < 1e-9 * max(1.0, np.abs(gpu_ctx.junction_get_estimates(red0)).max())
        D = np.eye(nk) * np.abs(np.diag(S)).mean() * (0.05 + 0.02 * rep)
        kk.upload_packed(pack_lower(S + D), nk)
        gpu_ctx.partial_complete_factor(pf, kk)
        gpu_ctx.partial_solve(0, pf)
        x_factor = gpu_ctx.block_get_corrections(0, ns)
        M = N0.copy()
        M[np.ix_(rws, rws)] += D
        ref = np.linalg.inv(M)
        x_ref = (ref @ rhs).reshape(x_factor.shape)
        scale = max(1e-30, np.abs(x_ref).max())
        assert np.abs(x_factor - x_ref).max() < 1e-9 * scale, np.abs(x_factor - x_ref).max() / scale
    # round 5 (a.reuse_factors): the light form reduces a right-hand side as well -- the forward half of the blocked substitution with the
    # kept factor gives what the elimination's passenger row gave: the reduced right-hand side of the complement
    r_elim = gpu_ctx.junction_get_estimates(red0).copy()
    gpu_ctx.junction_put_estimates(red, np.zeros(nk))
    gpu_ctx.partial_reduce_rhs(0, pf, red)
    gpu_ctx.sync()
    r_subst = gpu_ctx.junction_get_estimates(red)
    assert np.abs(r_subst - r_elim).max() < 1e-9 * max(1.0, np.abs(r_elim).max())
    gpu_ctx.partial_finish(pf, store, n0)                                   # the lender receives the inverse
    got = unpack_lower(store.download_packed(), n0)
    assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    with pytest.raises(Exception):
        gpu_ctx.partial_finish(pf, store, n0)
    with pytest.raises(Exception):
        gpu_ctx.partial_reduce_rhs(0, pf, red)                              # (the factor has become the inverse)
    gpu_ctx.partial_destroy(pf)
    for q in (m, red0, red, kk, store):
        q.close()
    gpu_ctx.block_destroy(0)


@pytest.mark.parametrize("rows,cols,pick,spine", [(9, 8, "jsl", True), (40, 30, "jsl", True), (40, 30, "scattered", False), (43, 43, "one", True)])
def test_normals_formed_in_elimination_order(gpu_ctx, built, orc, tmp_path, rows, cols, pick, spine):
    """dnagpu_block_form_reduce against dnagpu_form_normals + dnagpu_add_diag3x3 + dnagpu_block_reduce: the same condensed block, reduced
    right-hand side, solution and final inverse, bit for bit -- the same terms summed in the same order, only written somewhere else"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, 2, seed=rows)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, _, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    ns = len(st0)
    n0 = 3 * ns
    loc0 = {int(s): i for i, s in enumerate(st0)}
    stn = {"jsl": [loc0[int(s)] for s in net.jsl[net.jsl_off[0]:net.jsl_off[1]]], "scattered": list(range(1, ns, 7))[::-1], "one": [ns // 2]}[pick]
    idx = np.array(stn, dtype=np.uint32)
    nk = 3 * len(idx)
    rng = np.random.default_rng(rows)
    con_stn = np.array(sorted(rng.choice(ns, size=min(5, ns), replace=False)), dtype=np.uint32)
    con_w9 = np.concatenate([(lambda g: (g @ g.T + np.eye(3)).ravel())(rng.standard_normal((3, 3))) * 1e3 for _ in con_stn])
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    m = gpu_ctx.matrix(n0)
    kk = gpu_ctx.matrix(nk)
    outs = []
    for ordered in (False, True):
        red = gpu_ctx.matrix(nk)
        store = gpu_ctx.matrix(n0 + 256)
        pf = gpu_ctx.partial_create_spine(n0, nk, store) if spine else gpu_ctx.partial_create_in(n0, nk, store)
        if ordered:
            gpu_ctx.block_form_reduce(0, con_stn, con_w9, idx, red, pf)
        else:
            gpu_ctx.form_normals(0, m, ns)
            gpu_ctx.add_diag3x3(m, con_stn, con_w9)
            gpu_ctx.block_reduce(0, m, idx, red, keep=pf)
        S = red.download_packed().copy()
        r = gpu_ctx.junction_get_estimates(red).copy()
        kk.upload_packed(S, nk)
        gpu_ctx.partial_complete_factor(pf, kk)
        gpu_ctx.partial_solve(0, pf)
        x = gpu_ctx.block_get_corrections(0, ns).copy()
        gpu_ctx.partial_finish(pf, store, n0)
        outs.append((S, r, x, store.download_packed().copy()))
        gpu_ctx.partial_destroy(pf)
        store.close()
        red.close()
    for u, v in zip(outs[0], outs[1]):
        assert np.array_equal(u, v)
    for q in (m, kk):
        q.close()
    gpu_ctx.block_destroy(0)


@pytest.mark.parametrize("rows,cols,pick,lend", [(9, 8, "jsl", False), (40, 30, "jsl", True), (40, 30, "scattered", False), (43, 43, "one", True)])
def test_completion_in_two_halves(gpu_ctx, built, orc, tmp_path, rows, cols, pick, lend):
    """dnagpu_partial_complete_factor + dnagpu_partial_solve + dnagpu_partial_finish: the solution of an iteration from the completed
    factor (two triangular matrix-vector products) equals inverse x right-hand side, and the inverse formed afterwards is the one
    dnagpu_partial_complete gives -- bit for bit (same launches in the same order)"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, 2, seed=rows)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, _, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    ns = len(st0)
    n0 = 3 * ns
    loc0 = {int(s): i for i, s in enumerate(st0)}
    stn = {"jsl": [loc0[int(s)] for s in net.jsl[net.jsl_off[0]:net.jsl_off[1]]], "scattered": list(range(1, ns, 7))[::-1], "one": [ns // 2]}[pick]
    idx = np.array(stn, dtype=np.uint32)
    nk = 3 * len(idx)
    rws = (3 * idx[:, None] + np.arange(3)).ravel()
    N0 = unpack_lower(a.block_normals(0), n0)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    m = gpu_ctx.matrix(n0)
    red, kk = gpu_ctx.matrix(nk), gpu_ctx.matrix(nk)
    out = []
    for halves in (False, True):
        store = gpu_ctx.matrix(n0 + 256)
        inv = store if lend else gpu_ctx.matrix(n0)
        pf = gpu_ctx.partial_create_in(n0, nk, store) if lend else gpu_ctx.partial_create(n0, nk)
        m.upload_packed(a.block_normals(0), n0)
        gpu_ctx.block_reduce(0, m, idx, red, keep=pf)
        S = unpack_lower(red.download_packed(), nk)
        D = np.eye(nk) * np.abs(np.diag(S)).mean() * 0.05
        kk.upload_packed(pack_lower(S + D), nk)
        if halves:
            gpu_ctx.partial_complete_factor(pf, kk)
            gpu_ctx.partial_solve(0, pf)
            x_factor = gpu_ctx.block_get_corrections(0, ns)
            with pytest.raises(Exception):
                gpu_ctx.partial_complete_factor(pf, kk)            # the reduce's state is consumed
            gpu_ctx.partial_finish(pf, inv, n0)
            with pytest.raises(Exception):
                gpu_ctx.partial_finish(pf, inv, n0)                # ... and so is the factor
        else:
            gpu_ctx.partial_complete(pf, kk, inv, n0)
        out.append(inv.download_packed().copy())
        if halves:
            gpu_ctx.solve_corrections(0, inv)
            gpu_ctx.sync()
            x_inverse = gpu_ctx.block_get_corrections(0, ns)
            scale = max(1e-30, np.abs(x_inverse).max())
            assert np.abs(x_factor - x_inverse).max() < 1e-11 * scale, np.abs(x_factor - x_inverse).max() / scale
            M = N0.copy()
            M[np.ix_(rws, rws)] += D
            ref = np.linalg.inv(M)
            assert np.abs(unpack_lower(out[-1], n0) - ref).max() < 1e-9 * np.abs(ref).max()
        gpu_ctx.partial_destroy(pf)
        store.close()
        if not lend:
            inv.close()
    assert np.array_equal(out[0], out[1])
    for q in (m, red, kk):
        q.close()
    gpu_ctx.block_destroy(0)


@pytest.mark.parametrize("rows,cols,pick", [(9, 8, "jsl"), (40, 30, "jsl"), (40, 30, "scattered"), (43, 43, "one"), (12, 11, "all")])
def test_reduce_keep_and_complete(gpu_ctx, built, orc, tmp_path, rows, cols, pick):
    """dnagpu_block_reduce with a retained factor + dnagpu_partial_complete: the kept block is changed (what the junction
    chains add) and the full inverse of the changed matrix comes back in natural order -- against numpy, and the reduced
    system against the plain reduce"""
    from dynadjust_amd import adjust
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, 2, seed=rows)
    net = orc.Network(str(tmp_path / "s"), True)
    a = orc.Adjustment(net, True)
    a.prepare()
    st0, cml0, _, _ = _upload_block(gpu_ctx, net, a, 0, blk_id=0)
    ns = len(st0)
    n0 = 3 * ns
    loc0 = {int(s): i for i, s in enumerate(st0)}
    stn = {"jsl": [loc0[int(s)] for s in net.jsl[net.jsl_off[0]:net.jsl_off[1]]], "scattered": list(range(1, ns, 7))[::-1],
           "one": [ns // 2], "all": list(range(ns))}[pick]
    idx = np.array(stn, dtype=np.uint32)
    rws = (3 * idx[:, None] + np.arange(3)).ravel()
    N0 = unpack_lower(a.block_normals(0), n0)
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    m = gpu_ctx.matrix(n0)
    red0, red1 = gpu_ctx.matrix(3 * len(idx)), gpu_ctx.matrix(3 * len(idx))
    m.upload_packed(a.block_normals(0), n0)
    gpu_ctx.block_reduce(0, m, idx, red0)
    pf = gpu_ctx.partial_create(n0, 3 * len(idx))
    m.upload_packed(a.block_normals(0), n0)
    gpu_ctx.block_reduce(0, m, idx, red1, keep=pf)
    S0 = unpack_lower(red0.download_packed(), 3 * len(idx))
    S1 = unpack_lower(red1.download_packed(), 3 * len(idx))
    assert np.abs(S0 - S1).max() < 1e-10 * np.abs(S0).max()
    assert np.abs(gpu_ctx.junction_get_estimates(red0) - gpu_ctx.junction_get_estimates(red1)).max() < 1e-9 * max(1.0, np.abs(gpu_ctx.junction_get_estimates(red0)).max())
    # what the chains do to the kept block: a symmetric positive update
    rng = np.random.default_rng(rows)
    G = rng.standard_normal((3 * len(idx), 3 * len(idx)))
    D = G @ G.T * np.abs(np.diag(S1)).mean() * 0.1
    kk = gpu_ctx.matrix(3 * len(idx))
    kk.upload_packed(pack_lower(S1 + D), 3 * len(idx))
    inv = gpu_ctx.matrix(n0)
    gpu_ctx.partial_complete(pf, kk, inv, n0)
    got = unpack_lower(inv.download_packed(), n0)
    M = N0.copy()
    M[np.ix_(rws, rws)] += D
    ref = np.linalg.inv(M)
    assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    # the retained state is consumed
    with pytest.raises(Exception):
        gpu_ctx.partial_complete(pf, kk, inv, n0)
    # ... but the factor of the eliminated part still reduces new right-hand sides (next iteration of a GNSS-only network)
    x = gpu_ctx.block_get_stations(0, 1, ns)
    gpu_ctx.block_put_stations(0, 1, x + 0.01 * rng.standard_normal(x.shape))
    gpu_ctx.block_compute_b(0)
    gpu_ctx.form_rhs(0)
    gpu_ctx.partial_reduce_rhs(0, pf, red1)
    m.upload_packed(a.block_normals(0), n0)
    gpu_ctx.block_reduce(0, m, idx, red0)
    r_new, r_ref = gpu_ctx.junction_get_estimates(red1), gpu_ctx.junction_get_estimates(red0)
    assert np.abs(r_new - r_ref).max() < 1e-9 * max(1.0, np.abs(r_ref).max()), np.abs(r_new - r_ref).max()
    assert np.array_equal(unpack_lower(red1.download_packed(), 3 * len(idx)), S1)          # the complement is left alone
    gpu_ctx.partial_destroy(pf)
    for q in (m, red0, red1, kk, inv):
        q.close()
    gpu_ctx.block_destroy(0)


def test_chain_plan_steps_against_numpy(gpu_ctx):
    """dnagpu_chain_plan_*: chain steps as data.  A step adds a reduced system (all stations) and an information-form junction (some
    stations, linearised at other estimates) into one system, adds constraint blocks and eliminates all stations but the kept ones: complement,
    reduced right-hand side and the kept stations' estimates against dense numpy -- one step alone, three steps of unequal size as ONE
    batch (the largest member's padded shape), a step that eliminates nothing, then right-hand sides only through the kept factors."""
    rng = np.random.default_rng(7)

    def spd(n):
        a = rng.standard_normal((n, n))
        return a @ a.T + n * np.eye(n)

    n_blk = 40
    xyz = rng.standard_normal(3 * n_blk) * 1e3
    gpu_ctx.block_create(0, n_blk, 0)
    gpu_ctx.block_set_stations(0, xyz)
    cases, steps, keep_mats = [], [], []
    for n_stn, k_j, n_keep, n_con in ((30, 8, 10, 2), (30, 8, 10, 2), (22, 5, 7, 0), (30, 30, 12, 3), (9, 4, 9, 1)):
        est_idx = rng.permutation(n_blk)[:n_stn].astype(np.uint32)
        A, a_rhs = spd(3 * n_stn), rng.standard_normal(3 * n_stn)
        pos = rng.permutation(n_stn)[:k_j].astype(np.uint32)
        S, s_rhs = spd(3 * k_j), rng.standard_normal(3 * k_j)
        x = xyz.reshape(-1, 3)[est_idx].ravel()
        jest = x.reshape(-1, 3)[pos].ravel() + rng.standard_normal(3 * k_j) * 1e-3
        con_stn = rng.permutation(n_stn)[:n_con].astype(np.uint32)
        con_w9 = np.concatenate([spd(3).T.ravel() for _ in range(n_con)]) if n_con else np.zeros(0)
        keep = rng.permutation(n_stn)[:n_keep].astype(np.uint32)
        mA, mJ, out = gpu_ctx.matrix(3 * n_stn), gpu_ctx.matrix(3 * k_j), gpu_ctx.matrix(3 * n_keep)
        gpu_ctx.junction_payload_put(mA, A, a_rhs)
        gpu_ctx.junction_payload_put(mJ, S, jest, s_rhs)
        keep_mats += [mA, mJ, out]
        steps.append({"n_stn": n_stn, "est": (np.zeros(n_stn, dtype=np.uint32), est_idx), "sources": [(mA, 0, np.arange(n_stn, dtype=np.uint32)), (mJ, 1, pos)],
                      "con": (con_stn, con_w9) if n_con else None, "keep": keep, "out": out, "out_junction": 1})
        cases.append((n_stn, A, a_rhs, pos, S, s_rhs, x, jest, con_stn, con_w9, keep, mA, out))

    def expected(c, a_rhs=None):
        n_stn, A, a0, pos, S, s_rhs, x, jest, con_stn, con_w9, keep, mA, out = c
        a_rhs = a0 if a_rhs is None else a_rhs
        K, r = A.copy(), a_rhs.copy()
        rows = (3 * pos[:, None] + np.arange(3)).ravel()
        K[np.ix_(rows, rows)] += S
        r[rows] += s_rhs + S @ (jest - x[rows])
        for q, s in enumerate(con_stn):
            K[3 * s:3 * s + 3, 3 * s:3 * s + 3] += con_w9[9 * q:9 * q + 9].reshape(3, 3).T
        kr = (3 * keep[:, None] + np.arange(3)).ravel()
        ir = np.setdiff1d(np.arange(3 * n_stn), kr)
        if ir.size == 0:
            return K[np.ix_(kr, kr)], r[kr], x[kr]
        Kii = np.linalg.inv(K[np.ix_(ir, ir)])
        return K[np.ix_(kr, kr)] - K[np.ix_(kr, ir)] @ Kii @ K[np.ix_(ir, kr)], r[kr] - K[np.ix_(kr, ir)] @ Kii @ r[ir], x[kr]

    plan = gpu_ctx.chain_plan_create(steps, [0, 1, 4, 5])
    for batch in range(3):
        gpu_ctx.chain_plan_run(plan, batch)
    for c in cases:
        S_e, r_e, x_e = expected(c)
        F, est, rhs = gpu_ctx.junction_payload_get(c[-1], 3 * len(c[10]))
        assert rhs is not None and np.array_equal(est, x_e)
        assert np.abs(F - S_e).max() <= 1e-11 * np.abs(S_e).max() and np.abs(F - F.T).max() == 0.0
        assert np.abs(rhs - r_e).max() <= 1e-10 * max(1.0, np.abs(r_e).max())
    # new right-hand sides of the reduced systems: through the kept factors, all five steps in one launch
    new = []
    for c in cases:
        a2 = rng.standard_normal(3 * c[0])
        gpu_ctx.junction_payload_put(c[11], c[1], a2)
        new.append(a2)
    gpu_ctx.chain_plan_run_rhs(plan, 0, 3)
    gpu_ctx.sync()
    for c, a2 in zip(cases, new):
        S_e, r_e, x_e = expected(c, a2)
        F, est, rhs = gpu_ctx.junction_payload_get(c[-1], 3 * len(c[10]))
        assert np.abs(F - S_e).max() <= 1e-11 * np.abs(S_e).max()
        assert np.abs(rhs - r_e).max() <= 1e-10 * max(1.0, np.abs(r_e).max())
    gpu_ctx.chain_plan_destroy(plan)
    for m in keep_mats:
        m.close()
    gpu_ctx.block_destroy(0)
