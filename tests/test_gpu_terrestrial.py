"""Device path against the CPU oracle on mixed terrestrial + GNSS networks (types A B K C E M S V Z L H R I J P Q and direction sets D):
coordinates within 1e-8 m... of a NON-linear problem iterated by both sides with the same rules, variances within 1e-8
relative, statistics, geodetic station records; one chain, two chains, phased and simultaneous."""
import numpy as np
import pytest

from dynadjust_amd import adjust
from dynadjust_amd.device import unpack_lower
from tests import dnaformats as F
from tests import terrestrial_net as T

pytestmark = pytest.mark.gpu


def _device_run(folder, name, phased, **kw):
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode if phased else adjust.SimultaneousMode, **kw)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    return a, a.AdjustNetwork()


def _oracle_run(orc, base, phased):
    net = orc.Network(base, phased)
    o = orc.Adjustment(net, phased)
    o.prepare()
    return net, o, o.run()


def _compare(a, st, o, ost, tol_x=1e-8, tol_v=1e-8, row_noise=0.0):
    assert st == ost and a.CurrentIteration() == o.iterations()
    for i in range(o.iterations()):
        # (a correction computed through rows with relative noise e is off by up to e times its size)
        assert abs(a.GetIterationCorrection(i + 1) - o.max_correction(i + 1)) < 1e-8 + row_noise * abs(o.max_correction(i + 1))
    for b in range(a.blockCount()):
        assert np.array_equal(a.block_stations(b), o.block_stations(b))
        assert np.abs(a.block_estimates(b) - o.block_estimates(b)).max() < tol_x
        vo = o.block_variances(b)
        assert np.abs(a.block_variances_packed(b) - vo).max() / np.abs(vo).max() < tol_v


@pytest.mark.parametrize("rows,cols,blocks,phased,mt,types", [
    (5, 4, 1, False, False, "SL"),
    (5, 4, 1, False, False, "SVZLHRBKACEM"),
    (6, 5, 3, True, False, "SVZLHRBKACEMIJPQ"),
    (5, 5, 2, True, False, "SLIJPQ"),
    (5, 4, 1, False, False, "SLD"),
    (6, 5, 3, True, False, "SLVD"),
    (8, 5, 4, True, True, "SVZLHRBKACEMD"),
    (8, 5, 4, True, True, "SVZLHRBKACEMIJPQ"),
])
def test_terrestrial_parity_with_oracle(built, orc, tmp_path, rows, cols, blocks, phased, mt, types):
    b, (bst, bms) = T.build_mixed_network(str(tmp_path / "t"), rows, cols, blocks, seed=rows + blocks, types=types)
    net, o, ost = _oracle_run(orc, str(tmp_path / "t"), phased)
    a, st = _device_run(str(tmp_path), "t", phased, multi_thread=mt)
    assert st == 0 and a.CurrentIteration() >= 2
    # I / P: the reference's design row is a forward difference of the latitude over 0.1 mm (PartialD_Latitude,
    # dnatemplategeodesyfuncs.hpp:282): two latitudes 1.5e-11 rad apart, each good to one ulp (1.1e-16), i.e. 1e-5 relative
    # noise in the row that depends on the last bit of atan().  Device and host libm differ there, as two builds of the
    # reference would: those networks agree to the row noise, the others to 1e-8
    noisy = any(t in types for t in "IP")
    _compare(a, st, o, ost, tol_x=1e-7 if noisy else 1e-8, tol_v=1e-5 if noisy else 1e-8, row_noise=1e-5 if noisy else 0.0)
    assert np.abs(a.adjusted_coordinates(len(bst)) - b.truth).max() < 0.03
    # statistics
    a.GenerateStatistics()
    so, fo = o.statistics()
    to = o.tmsr_fields()
    assert a.GetMeasurementCount() == so.measurement_params and a.GetDegreesOfFreedom() == so.dof
    assert abs(a.GetChiSquared() - so.chi_squared) < (1e-5 if noisy else 1e-7) * so.chi_squared
    assert abs(a.GetGlobalPelzerRel() - so.global_pelzer) < (1e-4 if noisy else 1e-6)
    assert a.GetPotentialOutlierCount() == so.potential_outliers
    rec = np.frombuffer(a.measurement_records().tobytes(), dtype=F.MEASUREMENT_DT)
    tr = rec[net.t_record]
    # (residuals inherit the ulp of 4e6 m coordinates, 9.3e-10 m, a few times over)
    assert np.abs(tr["measCorr"] - to["measCorr"]).max() < 2e-8
    assert np.abs(tr["measAdj"] - to["measAdj"]).max() < 2e-8
    assert np.abs(tr["preAdjCorr"] - to["preAdjCorr"]).max() < 1e-12
    ptol = 2e-5 if noisy else 1e-7
    assert np.abs(tr["measAdjPrec"] - to["measAdjPrec"]).max() < ptol * np.abs(to["measAdjPrec"]).max()
    assert np.abs(tr["NStat"] - to["NStat"]).max() < (1e-3 if noisy else 1e-4)
    for k in range(a.blockCount()):
        po = o.block_prec_adj_msrs(k)
        assert np.abs(a.block_prec_adj_msrs(k) - po).max() < ptol * np.abs(po).max()
    a.close()
    o.close()


@pytest.mark.parametrize("mt,runs,types", [(True, 3, "SVZL"), (False, 4, "SLVD")])
def test_terrestrial_chains_in_lock_step(built, orc, tmp_path, mt, runs, types):
    """a.chain_runs on a network with terrestrial measurements: the normals move with the estimates, so every iteration factors again --
    the lock-step chains (dna_adjust::LockstepChains) run in their eliminating form in EVERY iteration (no kept factors, a.reuse_factors
    does not apply).  Twelve strips; against the oracle."""
    b, (bst, bms) = T.build_mixed_network(str(tmp_path / "t"), 26, 4, 12, seed=31, types=types)
    net, o, ost = _oracle_run(orc, str(tmp_path / "t"), True)
    a, st = _device_run(str(tmp_path), "t", True, multi_thread=mt, chain_runs=runs)
    assert st == 0 and a.CurrentIteration() >= 2 and a.blockCount() == 12
    assert a.chain_runs() == runs and a.chain_step_reuses() == 0
    _compare(a, st, o, ost, tol_x=1e-8, tol_v=1e-8)
    r, st2 = _device_run(str(tmp_path), "t", True, multi_thread=mt, chain_runs=0)
    assert r.chain_runs() == 0 and st2 == st and r.CurrentIteration() == a.CurrentIteration()
    for k in range(12):
        assert np.abs(r.block_estimates(k) - a.block_estimates(k)).max() < 1e-8
    a.close()
    r.close()
    o.close()


@pytest.mark.parametrize("phased,blocks", [(False, 1), (True, 3)])
def test_direction_sets_and_coordinates_from_text_files(built, orc, tmp_path, phased, blocks):
    """the measurement types no reference sample contains -- direction sets with ignored directions, I / J / P / Q -- from DNA text through
    the PRODUCT's importer (host/dnaimport_lite.cpp) to the device, against the oracle on the same imported files, and against the truth
    the synthetic observations were drawn from"""
    import shutil
    from tests.test_import import _write_dna_text
    types = "SVZLHRBCEMDJQ"            # (no A / K: the synthetic ones carry instrument heights the DNA format has no columns for;
    b, (bst, bms) = T.build_mixed_network(str(tmp_path / "t"), 6, 5, blocks, seed=7, types=types)     # no I / P: their row noise, see above)
    assert (bms["measType"] == b"D").sum() > 20 and ((bms["measType"] == b"D") & (bms["ignore"] != 0)).any()
    _write_dna_text(bst, bms, str(tmp_path / "t.stn"), str(tmp_path / "t.msr"))
    s = adjust.import_dna_text(str(tmp_path / "t.stn"), str(tmp_path / "t.msr"), str(tmp_path / "p"))
    assert s["records"] == len(bms)
    shutil.copy(str(tmp_path / "t.seg"), str(tmp_path / "p.seg"))       # same records in the same order: the segmentation carries over
    # geoid and deflections come from a .geo file in a real import; here from the synthetic station records
    pb = F.read_bst(str(tmp_path / "p.bst")).copy()
    for f in ("geoidSep", "verticalDef", "meridianDef"):
        pb[f] = bst[f]
    F.write_bst(str(tmp_path / "p.bst"), pb)
    net, o, ost = _oracle_run(orc, str(tmp_path / "p"), phased)
    a, st = _device_run(str(tmp_path), "p", phased)
    assert st == 0 and a.CurrentIteration() >= 2
    _compare(a, st, o, ost)
    assert np.abs(a.adjusted_coordinates(len(bst)) - b.truth).max() < 0.03
    a.close()
    o.close()


def test_reuse_inverses_is_ignored_for_non_gps_networks(built, tmp_path):
    """the design of terrestrial measurements follows the estimates: inverses cannot be kept"""
    T.build_mixed_network(str(tmp_path / "r"), 6, 4, 3, seed=2, types="SVZL")
    a, st = _device_run(str(tmp_path), "r", True, reuse_inverses=True)
    B = a.blockCount()
    assert st == 0 and a.solve_count() == a.CurrentIteration() * (3 * B - 2)
    a.close()


@pytest.mark.parametrize("phased,mt,sample,blocks", [(False, False, "gda94", 2), (True, False, "gda94", 2), (True, True, "gda94", 2),
                                                     (True, True, "gda2020", 3)])
def test_reference_urban_sample_on_the_device(built, golden_dir, tmp_path, phased, mt, sample, blocks):
    """The reference's urban sample (248 angles, 427 slope distances, 287 zenith distances, levelling, azimuths, GNSS
    baselines and an LLH point cluster, mixed station constraints, deflections, geoid) through the facade, the way the
    published report was made: adjust, UpdateBinaryFiles, adjust again from the updated (reduced) files -- against
    urban.phased.adj.expected: summary figures, all 149 adjusted positions with their standard deviations, and the full
    adjusted-measurement table (1182 rows: adjusted value, correction, precision, N-stat, Pelzer, pre-adjustment correction)"""
    from tests import urban_net as U
    from tests.test_oracle_terrestrial import _check_urban_tables
    from tests.dnatext import cart_to_geo
    from dynadjust_amd.device import unpack_lower
    base = str(tmp_path / "urban")
    # (gda2020: urban_mt.phased-mt.adj.expected, the reference's multi-thread run on the network after dnareftran, 3 blocks)
    stations, msrs, rep, bst, bms, first_of = U.build_urban_sample(golden_dir, base, blocks=blocks, sample=sample)
    for run in range(2):
        a, st = _device_run(str(tmp_path), "urban", phased, multi_thread=mt)
        assert st == 0
        a.GenerateStatistics()
        if run == 0:
            a.UpdateBinaryFiles()
            a.close()
    assert a.GetMeasurementCount() == rep["measurements"] and a.GetUnknownsCount() == rep["unknowns"] and a.GetDegreesOfFreedom() == rep["dof"]
    assert abs(a.GetChiSquared() - rep["chi2"]) < 0.2 and abs(a.GetSigmaZero() - rep["sigma0"]) < 8e-4
    assert abs(a.GetGlobalPelzerRel() - rep["pelzer"]) < 6e-4 and a.GetPotentialOutlierCount() == rep["outliers"]
    assert a.CurrentIteration() == 1                      # like the report: nothing moves any more
    bs = [a.block_stations(k) for k in range(a.blockCount())]
    be = [a.block_estimates(k) for k in range(a.blockCount())]
    sd = []
    for k in range(a.blockCount()):
        V = unpack_lower(a.block_variances_packed(k), 3 * len(bs[k]))
        out = np.zeros((len(bs[k]), 3))
        for p in range(len(bs[k])):
            lat, lon, _ = cart_to_geo(*be[k][3 * p:3 * p + 3])
            R = np.array([[-np.sin(lon), np.cos(lon), 0.0],
                          [-np.sin(lat) * np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat)],
                          [np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)]])
            out[p] = np.sqrt(np.diag(R @ V[3 * p:3 * p + 3, 3 * p:3 * p + 3] @ R.T))
        sd.append(out)
    rec = np.frombuffer(a.measurement_records().tobytes(), dtype=F.MEASUREMENT_DT)
    names = ("measAdj", "measCorr", "measAdjPrec", "NStat", "PelzerRel", "preAdjCorr")
    t_record = [first_of[c] for c, m in enumerate(msrs) if m["type"] not in "GXY" and not m["ignore"]]
    tf = {nm: rec[nm][t_record] for nm in names}
    vec_of_record, rows3 = {}, []
    for c, m in enumerate(msrs):
        if m["type"] not in "GXY" or m["ignore"]:
            continue
        vec_of_record[first_of[c]] = len(rows3) // 3
        q = first_of[c]
        for j in range(len(m["vectors"])):
            rows3 += [q, q + 1, q + 2]
            q += 3 + 3 * int(rec["vectorCount2"][q]) * (m["type"] != "G")
    gf = {nm: rec[nm][rows3] for nm in names}
    _check_urban_tables(rep, stations, msrs, first_of, t_record, bs, be, sd, tf, gf, vec_of_record, loose=6.0 if sample == "gda2020" else 1.0)
    a.close()


@pytest.mark.parametrize("phased,kw", [(False, {}), (True, {"schur_carry": False}), (True, {"schur_carry": True, "multi_thread": True})])
def test_oscillation_diagnostics_on_the_device(built, orc, tmp_path, phased, kw):
    """dna_adjust::UpdateIterationDiagnostics / PrintOscillationSummary / PrintSuspectMeasurementSummary (ADJ:7450-7780) behind the reference's
    members: the corrections of every iteration are compared on the device (osc_update_stations_kernel: one launch for all blocks, a thread takes a station through its visits in block order; one record per station of the network), the host
    keeps the history.  On a network that cannot settle (two stations held across their lines by two distances too short to meet) the device
    records the same stations, iterations and cycle counts as the oracle's restatement, the same magnitudes (the iteration is chaotic: rounding
    differences double every iteration -- 1e-6 relative after ten), the last correction rotated into the station's local frame, and prints the
    reference's summary lines -- under simultaneous adjustment, the reference's phased schedule and the condensed schedule."""
    base = str(tmp_path / "o")
    T.build_oscillating_network(base, blocks=2 if phased else 1)
    net = orc.Network(base, phased)
    o = orc.Adjustment(net, phased, max_iterations=10)
    o.prepare()
    ost = o.run()
    want = o.oscillation_history()
    assert ost == adjust.ADJUST_MAX_ITERATIONS_EXCEEDED and [r["station"] for r in want] == [1, 7]
    a, st = _device_run(str(tmp_path), "o", phased, max_iterations=10, **kw)
    assert st == ost and a.CurrentIteration() == 10
    got = a.oscillation_history()
    assert [r["station"] for r in got] == [1, 7]
    bst = F.read_bst(base + ".bst")
    for g, w in zip(got, want):
        for k in ("first_iteration", "last_iteration", "cycles"):
            assert g[k] == w[k], (k, g, w)
        assert abs(g["first_mag"] - w["first_mag"]) < 1e-6 * w["first_mag"] and abs(g["last_mag"] - w["last_mag"]) < 1e-6 * w["last_mag"]
        e, n, u = T.enu_axes(float(bst["currentLatitude"][g["station"]]), float(bst["currentLongitude"][g["station"]]))
        back = g["last_e"] * e + g["last_n"] * n + g["last_up"] * u
        assert np.abs(back - np.array(w["last_xyz"])).max() < 1e-6 * w["last_mag"]
    a.GenerateStatistics()
    text = a.summaries(limit=5)
    assert "+ Oscillating stations detected (2 total, showing top 2):" in text
    big = max(want, key=lambda r: max(r["first_mag"], r["last_mag"]))
    line = [l for l in text.splitlines() if l.startswith("  - T%05d" % big["station"])]
    assert line and "%.1fm to %.1fm" % (big["first_mag"], big["last_mag"]) in line[0] and "%d cycles" % big["cycles"] in line[0]
    assert "(iterations %d-%d)" % (big["first_iteration"], big["last_iteration"]) in line[0] and " — " in line[0]
    assert "+ Suspect measurements connected to oscillating stations (" in text and "touches oscillating station" in text
    # an adjustment that converges records nothing and prints no oscillation summary
    a.close()
    T.build_mixed_network(str(tmp_path / "m"), rows=4, cols=3, blocks=2 if phased else 1, types="SVL")
    a, st = _device_run(str(tmp_path), "m", phased, **kw)
    assert st == 0 and a.oscillation_history() == [] and "Oscillating" not in a.summaries()
    a.close()
    o.close()
