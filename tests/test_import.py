"""The product-side importer (dnaimport_text, include/dnaadjust_c.h; host/dnaimport_lite.cpp): DNA text station / measurement files to
the binary files the adjustment reads, GNSS measurements aligned to the stations' frame like dnareftran does.  Host code only: runs
without a GPU.  Pinned on the reference's own sample: the transformed observations are the "Measured" column of the reference's
report (printed to 1e-4 m), which the raw .msr values miss by up to 1.8 mm."""
import os

import numpy as np
import pytest

from dynadjust_amd import adjust
from tests import dnaformats as F
from tests import dnatext as T


def test_sample_network_matches_the_reference_report(built, golden_dir, tmp_path):
    base = str(tmp_path / "net")
    s = adjust.import_dna_text(os.path.join(golden_dir, "gnss-network.stn"), os.path.join(golden_dir, "gnss-network.msr"), base)
    assert s == {"stations": 43, "records": 480, "vectors": 139, "clusters": 131, "vectors_transformed": 133}
    adj = T.read_adj(os.path.join(golden_dir, "gnss.simult.adj.expected"))
    bms = F.read_bms(base + ".bms")
    obs = bms["term1"][bms["measStart"] < 3]
    measured = np.array([m["measured"] for m in adj["msr"]])
    assert obs.size == measured.size == 417
    assert np.abs(obs - measured).max() < 0.5e-4 + 1e-9           # the report prints four decimals
    raw = np.array([x for c in T.read_msr(os.path.join(golden_dir, "gnss-network.msr")) for v in c["vectors"] for x in v[2]])
    assert np.abs(raw - measured).max() > 1.5e-3                  # the alignment matters: ITRF2008 @ 2015 / ITRF2014 @ 2018 -> GDA2020
    # same records as the test-only writer lays out (types, stations, cluster bookkeeping, variances, scalars)
    ref_base = str(tmp_path / "ref")
    stn, cl, _ = T.build_gnss_sample(golden_dir, ref_base)
    ref = F.read_bms(ref_base + ".bms")
    for f in ("measType", "measStart", "station1", "station2", "vectorCount1", "vectorCount2", "clusterID", "scale1", "scale2", "scale3", "scale4",
              "term2", "term3", "term4", "ignore"):
        assert np.array_equal(bms[f], ref[f]), f
    cov = bms["measStart"] >= 3
    assert np.array_equal(bms["term1"][cov], ref["term1"][cov])
    bst, rbst = F.read_bst(base + ".bst"), F.read_bst(ref_base + ".bst")
    for f in ("stationName", "stationConst"):
        assert np.array_equal(bst[f], rbst[f])
    for f in ("currentLatitude", "currentLongitude"):
        assert np.abs(bst[f] - rbst[f]).max() < 1e-12
    assert np.abs(bst["currentHeight"] - rbst["currentHeight"]).max() < 1e-5       # (XYZ stations: the reference's CartToGeo against an iterated one)
    assert np.array_equal(np.asarray(F.read_asl(base + ".asl")), np.asarray(F.read_asl(ref_base + ".asl")))


def test_urban_sample_equals_the_test_side_import(built, golden_dir, tmp_path):
    """the reference's urban sample (UTM stations, 1 112 measurements of types A B G H K L M S V Y Z, 17 ignored, a 4-point LLH
    cluster, the exported geoid file) through the product's importer: the same station and measurement records, field by field, as
    the test-side importer whose files the oracle and the device adjust onto the reference's urban.phased.adj.expected
    (tests/test_oracle_terrestrial.py, tests/test_gpu_terrestrial.py).  One documented difference: the geoid separation comes from
    the .geo file (3 decimals) here, refined to 4 decimals from the report's h - H columns there."""
    from tests import urban_net as U
    g = golden_dir
    s = adjust.import_dna_text(os.path.join(g, "urban-network.stn"), os.path.join(g, "urban-network.msr"), str(tmp_path / "p"),
                               os.path.join(g, "urban-network.geo"))
    assert s["stations"] == 149 and s["clusters"] == 1112 and s["vectors_transformed"] == 0
    stations, msrs, rep, bst, bms, first_of = U.build_urban_sample(g, str(tmp_path / "t"))
    pb, pm = F.read_bst(str(tmp_path / "p.bst")), F.read_bms(str(tmp_path / "p.bms"))
    assert len(pb) == len(bst) == 149 and len(pm) == len(bms)
    for f in ("stationName", "stationConst", "suppliedStationType"):
        assert np.array_equal(pb[f], bst[f]), f
    for f in ("currentLatitude", "currentLongitude", "meridianDef", "verticalDef"):
        assert np.abs(pb[f] - bst[f]).max() < 1e-14, f
    assert np.abs(pb["geoidSep"].astype(float) - bst["geoidSep"].astype(float)).max() < 5.1e-4          # .geo: 3 decimals
    assert np.abs((pb["currentHeight"] - pb["geoidSep"]) - (bst["currentHeight"] - bst["geoidSep"])).max() < 1e-6   # the supplied H
    for f in ("measType", "measStart", "ignore", "station1", "station2", "station3", "vectorCount1", "vectorCount2", "clusterID",
              "measurementStations", "term2", "term3", "term4", "scale4"):
        assert np.array_equal(pm[f], bms[f]), f
    for f in ("term1", "preAdjMeas"):
        assert np.abs(pm[f] - bms[f]).max() < 1e-14, f
    assert np.array_equal(np.asarray(F.read_asl(str(tmp_path / "p.asl"))), np.asarray(F.read_asl(str(tmp_path / "t.asl"))))


def _dms(rad, decimals=9):
    """radians -> ("-ddd", "mm", "ss.sssssssss") of the DNA measurement columns"""
    sgn = "-" if rad < 0 else ""
    a = abs(rad) * 180.0 / np.pi
    d = int(a)
    m = int((a - d) * 60.0)
    sec = ((a - d) * 60.0 - m) * 60.0
    if round(sec, decimals) >= 60.0:
        sec = 0.0
        m += 1
    if m >= 60:
        m -= 60
        d += 1
    return "%s%d" % (sgn, d), "%02d" % m, "%0*.*f" % (decimals + 3, decimals, sec)


def _ddmmss(rad):
    """radians -> ddd.mmsssss... of the DNA station columns"""
    d, m, sec = _dms(rad, 8)
    return "%s.%s%s" % (d, m, sec.replace(".", ""))


def _write_dna_text(bst, bms, stn_path, msr_path):
    """the binary network of tests/terrestrial_net.py as DNA v3.01 text files (what dnaimport would be given)"""
    name = lambda i: bst["stationName"][i].decode().rstrip("\0").strip()
    with open(stn_path, "w") as f:
        f.write("!#=DNA 3.01 STN    01.01.2020       GDA2020    01.01.2020 %9d\n" % len(bst))
        for i in range(len(bst)):
            f.write("%-20s%-3s %-3s%20s%20s%20.6f\n" % (name(i), bst["stationConst"][i].decode().rstrip("\0"), "LLh", _ddmmss(bst["currentLatitude"][i]),
                                                    _ddmmss(bst["currentLongitude"][i]), bst["currentHeight"][i]))
    sec = np.pi / 648000.0
    col = lambda t, ig, a="", b="", c="": "%s%s%-20s%-20s%-20s" % (t, "*" if ig else " ", a, b, c)
    with open(msr_path, "w") as f:
        f.write("!#=DNA 3.01 MSR    01.01.2020       GDA2020    01.01.2020 %9d\n" % len(bms))
        i = 0
        while i < len(bms):
            r = bms[i]
            t = r["measType"].decode()
            if t == "D":
                k = int(r["vectorCount1"])
                d, m, sx = _dms(r["term1"])
                f.write(col("D", r["ignore"], name(r["station1"]), name(r["station2"]), str(k - 1)) + " %s %s %s %.9f\n" % (d, m, sx, np.sqrt(r["term2"]) / sec))
                for q in bms[i + 1:i + k]:
                    d, m, sx = _dms(q["term1"])
                    f.write(col("D", q["ignore"], "", "", name(q["station2"])) + " %s %s %s %.9f\n" % (d, m, sx, np.sqrt(q["term2"]) / sec))
                i += k
            elif t in "GY":
                x, y, z = bms[i], bms[i + 1], bms[i + 2]
                if t == "G":
                    f.write(col("G", r["ignore"], name(r["station1"]), name(r["station2"])) + "\n")
                else:
                    f.write(col("Y", r["ignore"], name(r["station1"]), "XYZ", "1") + "\n")
                pad = " " * 62
                f.write(pad + " %.9f %.12e\n" % (x["term1"], x["term2"]))
                f.write(pad + " %.9f %.12e %.12e\n" % (y["term1"], y["term2"], y["term3"]))
                f.write(pad + " %.9f %.12e %.12e %.12e\n" % (z["term1"], z["term2"], z["term3"], z["term4"]))
                i += 3
            else:
                stn = [name(r["station1"])] + ([name(r["station2"])] if r["measurementStations"] >= 2 else []) + ([name(r["station3"])] if r["measurementStations"] >= 3 else [])
                if t in "ABIJKPQVZ":
                    d, m, sx = _dms(r["term1"])
                    val = "%s %s %s %.9f" % (d, m, sx, np.sqrt(r["term2"]) / sec)
                else:
                    val = "%.9f %.9f" % (r["term1"], np.sqrt(r["term2"]))
                if t in "SVZ":
                    val += " %.6f %.6f" % (r["term3"], r["term4"])
                f.write(col(t, r["ignore"], *stn) + " " + val + "\n")
                i += 1


def test_every_measurement_type_through_the_text_files(built, tmp_path):
    """a synthetic network with all 17 terrestrial types -- direction sets (with ignored directions) and the single-station I / J / P / Q
    included, which the reference's samples do not contain -- written as DNA text and imported by the product: the records the
    adjustment reads (types, stations, set bookkeeping, values, variances, instrument / target heights) equal those of the directly
    written binary network that the oracle and the device adjust (tests/test_gpu_terrestrial.py)"""
    from tests import terrestrial_net as TN
    b, (bst, bms) = TN.build_mixed_network(str(tmp_path / "t"), rows=6, cols=5, blocks=1, seed=4, types="SVZLHRBKACEMDIJPQ")
    types = set(x.decode() for x in bms["measType"])
    assert types == set("SVZLHRBKACEMDIJPQGY")
    assert (bms["ignore"] != 0).any()                                  # a dropped direction somewhere
    _write_dna_text(bst, bms, str(tmp_path / "t.stn"), str(tmp_path / "t.msr"))
    s = adjust.import_dna_text(str(tmp_path / "t.stn"), str(tmp_path / "t.msr"), str(tmp_path / "p"))
    assert s["stations"] == len(bst) and s["records"] == len(bms) and s["vectors_transformed"] == 0
    pb, pm = F.read_bst(str(tmp_path / "p.bst")), F.read_bms(str(tmp_path / "p.bms"))
    assert np.abs(pb["currentLatitude"] - bst["currentLatitude"]).max() < 1e-12 and np.abs(pb["currentLongitude"] - bst["currentLongitude"]).max() < 1e-12
    assert np.abs(pb["currentHeight"] - bst["currentHeight"]).max() < 1e-6
    for f in ("measType", "measStart", "station1", "station2", "station3", "ignore", "measurementStations"):
        assert np.array_equal(pm[f], bms[f]), f
    dset = bms["measType"] == b"D"
    for f in ("vectorCount1", "vectorCount2", "clusterID"):
        assert np.array_equal(pm[f][dset], bms[f][dset]), f
    ang = np.isin(bms["measType"], [x.encode() for x in "ABDIJKPQVZ"])
    assert np.abs(pm["term1"][ang] - bms["term1"][ang]).max() < 1e-13              # 1e-9 seconds of arc
    assert np.abs(pm["term1"][~ang] - bms["term1"][~ang]).max() < 1e-8
    heights = np.isin(bms["measType"], [b"S", b"V", b"Z"])                     # term3 / term4: instrument / target height, six decimals
    for f in ("term3", "term4"):
        assert np.abs(pm[f][heights] - bms[f][heights]).max() < 1e-6, f
    gnss = np.isin(bms["measType"], [b"G", b"Y"])                             # term3 / term4: covariances (the synthetic A / K records carry
    for f, rows in (("term2", np.ones(len(bms), bool)), ("term3", gnss), ("term4", gnss)):     # heights the DNA format has no columns for)
        a, e = pm[f][rows], bms[f][rows]
        assert np.array_equal(a == 0, e == 0), f
        nz = e != 0
        assert np.abs(a[nz] / e[nz] - 1.0).max() < 1e-6, f                    # variances: from standard deviations printed to 1e-9
    assert np.array_equal(np.asarray(F.read_asl(str(tmp_path / "p.asl"))), np.asarray(F.read_asl(str(tmp_path / "t.asl"))))


def test_utm_against_the_test_side_series(built):
    import ctypes as C
    from tests import urban_net as U
    lat, lon = U.utm_to_geo(320236.2750, 5813988.8399, 55)
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, "u.stn"), "w").write("!#=DNA 3.01 STN    01.01.2020       GDA2020    01.01.2020         1\n"
                                              "1                   FFF UTM         320236.2750        5813988.8399             31.4770 55 1   \n")
    open(os.path.join(d, "u.msr"), "w").write("!#=DNA 3.01 MSR    01.01.2020       GDA2020    01.01.2020         0\n")
    adjust.import_dna_text(os.path.join(d, "u.stn"), os.path.join(d, "u.msr"), os.path.join(d, "u"))
    bst = F.read_bst(os.path.join(d, "u.bst"))
    assert abs(bst["currentLatitude"][0] - lat) < 1e-14 and abs(bst["currentLongitude"][0] - lon) < 1e-14
    assert -0.67 < lat < -0.65 and 2.5 < lon < 2.55          # Melbourne


def test_frame_alignment_formulas(built):
    """decimal year and 14-parameter transformation against hand-computed values"""
    import ctypes as C
    # a baseline of 100 km rotates by the ITRF2014 -> GDA2020 plate rotation over 2 years: |d| ~ 100 km x 2.2 mas/yr x 2 yr ~ 2 mm
    stn = "A                   FFF XYZ       -4000000.0000        3000000.0000       -3000000.0000    \n" \
          "B                   FFF XYZ       -3950000.0000        3050000.0000       -3070000.0000    \n"
    msr = ("G A                   B                                             1.00      1.00      1.00      1.00            ITRF2014          01.01.2018\n"
           "                                                                        50000.0000 1.0e-06\n"
           "                                                                        50000.0000 0.0 1.0e-06\n"
           "                                                                       -70000.0000 0.0 0.0 1.0e-06\n")
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, "a.stn"), "w").write("!#=DNA 3.01 STN    01.01.2020       GDA2020    01.01.2020         2\n" + stn)
    open(os.path.join(d, "a.msr"), "w").write("!#=DNA 3.01 MSR    01.01.2020       GDA2020    01.01.2020         1\n" + msr)
    adjust.import_dna_text(os.path.join(d, "a.stn"), os.path.join(d, "a.msr"), os.path.join(d, "a"))
    bms = F.read_bms(os.path.join(d, "a.bms"))
    dt = (2018.0 + 0.5 / 365.0) - 2020.0
    mas = np.pi / 180 / 3600 / 1000
    rx, ry, rz = (np.array([1.50379, 1.18346, 1.20716]) * dt * mas)
    R = np.array([[1, rz, -ry], [-rz, 1, rx], [ry, -rx, 1]])
    exp = R @ np.array([50000.0, 50000.0, -70000.0])
    assert np.abs(bms["term1"][:3] - exp).max() < 1e-9
    assert 1e-3 < np.abs(exp - np.array([50000.0, 50000.0, -70000.0])).max() < 3e-3


@pytest.mark.parametrize("stn,msr,message", [
    ("A                   FFF ENU        500000.0000        6000000.0000            10.0000    \n", "", "not supported"),
    ("A                   FFF LLH      -36.3348253617      145.5741006771            172.1933    \n",
     "T A                   A                                                           100.0 0.01\n", "not supported"),
    ("A                   FFF LLH      -36.3348253617      145.5741006771            172.1933    \n",
     "D A                   A                                                           100 00 00.0 1.0\n", "without a direction count"),
    ("A                   FFF LLH      -36.3348253617      145.5741006771            172.1933    \n",
     "D A                   A                   2                    100 00 00.0 1.0\n"
     "D                                         A                    120 00 00.0 1.0\n", "cut short"),
    ("A                   FFF LLH      -36.3348253617      145.5741006771            172.1933    \n",
     "G A                   NOWHERE                                       1.00      1.00      1.00      1.00             GDA2020          01.01.2020\n"
     "   1.0 1e-6\n   1.0 0 1e-6\n   1.0 0 0 1e-6\n", "is not in the station file"),
    ("A                   FFF LLH      -36.3348253617      145.5741006771            172.1933    \n"
     "B                   FFF LLH      -36.3348253617      145.5841006771            172.1933    \n",
     "G A                   B                                             1.00      1.00      1.00      1.00               WGS84          01.01.2020\n"
     "                                                                  1.0 1e-6\n                                                                  1.0 0 1e-6\n"
     "                                                                  1.0 0 0 1e-6\n", "no transformation"),
])
def test_import_errors(built, tmp_path, stn, msr, message):
    open(tmp_path / "e.stn", "w").write("!#=DNA 3.01 STN    01.01.2020       GDA2020    01.01.2020         1\n" + stn)
    open(tmp_path / "e.msr", "w").write("!#=DNA 3.01 MSR    01.01.2020       GDA2020    01.01.2020         1\n" + msr)
    with pytest.raises(RuntimeError) as e:
        adjust.import_dna_text(str(tmp_path / "e.stn"), str(tmp_path / "e.msr"), str(tmp_path / "e"))
    assert message in str(e.value)
