"""CPU oracle: terrestrial measurement types (UpdateDesignNormalMeasMatrices_A/_BK/_CEM/_E/_M/_S/_V/_Z/_L/_H/_R,
dnaadjust.cpp:4754-6054).  The networks come from tests/terrestrial_net.py, whose measurement models are written
independently (numpy, local-frame vectors)."""
import numpy as np
import pytest

from dynadjust_amd.device import unpack_lower
from tests import terrestrial_net as T


def _run(orc, base, phased, **kw):
    net = orc.Network(base, phased)
    a = orc.Adjustment(net, phased, **kw)
    a.prepare()
    return net, a, a.run()


def test_design_rows_are_the_derivatives(orc, tmp_path):
    """every design row against central differences of the computed measurement (station geodetic data held fixed, as in
    the reference, where the bst latitude / longitude only change between iterations)"""
    b, (bst, bms) = T.build_mixed_network(str(tmp_path / "d"), 4, 4, 1, seed=4, types="SVZLHRBKACEMIJPQ")
    net, a, st = _run(orc, str(tmp_path / "d"), False, max_iterations=1)
    seen = set()
    for t in range(net.n_tmsr):
        ty = chr(net.t_type[t])
        stn = net.t_stn[3 * t:3 * t + 3]
        ns = 3 if ty == "A" else (1 if ty in "HRIJPQ" else 2)
        X = np.zeros(9)
        for q in range(ns):
            X[3 * q:3 * q + 3] = b.init[stn[q]]
        comp, row = a.tmsr_evaluate(t, X)
        for i in range(3 * ns):
            d = 1.0 if ty in "HRL" else 1e-3       # heights come out of sqrt(...) - nu: 1e-9 m of cancellation noise
            xp, xm = X.copy(), X.copy()
            xp[i] += d
            xm[i] -= d
            num = (a.tmsr_evaluate(t, xp)[0] - a.tmsr_evaluate(t, xm)[0]) / (xp[i] - xm[i])    # (the step as actually taken)
            # C / E / M: the reference's row is -d/|d| of the chord between the points REDUCED to the ellipsoid and leaves
            # out the reduction factor nu/(nu+h) ~ 1 - 3e-5 (dnaadjust.cpp:5068-5071): an approximate Jacobian by design
            rel = 1e-4 if ty in "CEM" else 2e-8
            if ty in "JQ":
                # the computed longitude is the station RECORD's (constant in X, Y, Z; dnaadjust.cpp:5936): the row is the
                # analytic d(longitude)/d(X, Y), checked against atan2 instead
                lon = lambda x: np.arctan2(x[1], x[0])
                num = (lon(xp) - lon(xm)) / (xp[i] - xm[i])
                rel = 1e-6
            if ty in "IP":
                rel = 1e-6                          # the reference's row is itself a forward difference with a 0.1 mm step
            assert abs(num - row[i]) < rel * max(1.0, abs(row[i])) + 2e-10, (ty, i, num, row[i])
        assert np.all(row[3 * ns:] == 0.0)
        seen.add(ty)
    assert seen == set("SVZLHRBKACEMIJPQ")
    a.close()


@pytest.mark.parametrize("types,defl,geoid", [("SL", False, False), ("SLHR", False, True), ("SVZ", True, False), ("SLBKA", True, False),
                                              ("CEMSL", False, True), ("SVZLHRBKACEM", True, True), ("SLPQ", False, False),
                                              ("SLIJ", True, False), ("SVZLHRBKACEMIJPQ", True, True)])
def test_adjustment_recovers_the_truth(orc, tmp_path, types, defl, geoid):
    b, _ = T.build_mixed_network(str(tmp_path / "n"), 5, 4, 1, seed=7, types=types, defl=defl, geoid=geoid)
    net, a, st = _run(orc, str(tmp_path / "n"), False)
    assert st == 0, (types, a.iterations())
    x = a.block_estimates(0).reshape(-1, 3)
    err = np.abs(x - b.truth).max()
    assert err < 0.02, (types, err)                      # millimetre-level observations, a few centimetres of initial error
    s, f = a.statistics()
    assert 0.4 < s.sigma_zero < 2.0, s.sigma_zero         # the stochastic model matches the simulated noise
    a.close()


@pytest.mark.parametrize("blocks", [2, 3])
def test_phased_is_rigorous_with_terrestrial_measurements(orc, tmp_path, blocks):
    b, _ = T.build_mixed_network(str(tmp_path / "p"), 6, 4, blocks, seed=11)
    ns, s, st_s = _run(orc, str(tmp_path / "p"), False)
    npn, p, st_p = _run(orc, str(tmp_path / "p"), True)
    assert st_s == 0 and st_p == 0
    xs = s.block_estimates(0).reshape(-1, 3)
    Vs = unpack_lower(s.block_variances(0), 3 * ns.n_stations)
    for k in range(p.n_blocks):
        stn = p.block_stations(k)
        assert np.abs(p.block_estimates(k).reshape(-1, 3) - xs[stn]).max() < 2e-6    # both iterate a non-linear problem to 0.5 mm
        idx = (3 * stn[:, None] + np.arange(3)).ravel()
        Vb = unpack_lower(p.block_variances(k), 3 * len(stn))
        assert np.abs(Vb - Vs[np.ix_(idx, idx)]).max() / np.abs(Vs).max() < 1e-4
    ss, _ = s.statistics()
    sp, _ = p.statistics()
    assert abs(ss.chi_squared - sp.chi_squared) < 1e-3 * ss.chi_squared and ss.dof == sp.dof
    s.close()
    p.close()
