"""Test-only reader of DNA-format text station / measurement files and of the measurement and coordinate tables of a
`.adj` report, enough to turn the reference's own sample network (sampleData/gnss-network.{stn,msr} with
gnss.simult.adj.expected, copied as data under tests/golden/) into .bst/.bms/.asl files and expected values.

The product never reads these formats: dnaimport is out of scope; this stands in for it in the tests.
GNSS types only (G baselines, X baseline clusters, Y point clusters in XYZ)."""
import math
import re

import numpy as np

from . import dnaformats as F

_NUM = re.compile(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?|[-+]?\d+(?:[eE][-+]?\d+)")


def dms_to_deg(v):
    """ddd.mmssssss -> decimal degrees (the DNA 'LLH' station notation)"""
    s = -1.0 if v < 0 else 1.0
    v = abs(v)
    d = math.floor(v + 1e-12)
    m = math.floor((v - d) * 100.0 + 1e-9)
    sec = ((v - d) * 100.0 - m) * 100.0
    return s * (d + m / 60.0 + sec / 3600.0)


def cart_to_geo(x, y, z):
    """GRS80, iterative; radians + ellipsoidal height"""
    a, f = 6378137.0, 1.0 / 298.257222101
    e2 = 2 * f - f * f
    lon = math.atan2(y, x)
    p = math.hypot(x, y)
    lat = math.atan2(z, p * (1 - e2))
    for _ in range(12):
        nu = a / math.sqrt(1 - e2 * math.sin(lat) ** 2)
        h = p / math.cos(lat) - nu
        lat = math.atan2(z, p * (1 - e2 * nu / (nu + h)))
    return lat, lon, h


def read_stn(path):
    """-> list of (name, constraint, lat_rad, lon_rad, height, description)"""
    out = []
    for line in open(path):
        if line.startswith("!#=") or line.startswith("*") or not line.strip():
            continue
        name = line[0:20].strip()
        const = line[20:23]
        ctype = line[24:27]
        vals = line[27:87].split()
        a, b, c = (float(v) for v in vals[:3])
        if ctype == "LLH":
            lat, lon, h = math.radians(dms_to_deg(a)), math.radians(dms_to_deg(b)), c
        elif ctype == "XYZ":
            lat, lon, h = cart_to_geo(a, b, c)
        else:
            raise ValueError("station coordinate type " + ctype)
        out.append((name, const, lat, lon, h, line[87:].strip()))
    return out


def read_msr(path):
    """-> list of clusters: dict(type, vscale, vectors=[(stn1, stn2, [x, y, z])], V = full symmetric 3k x 3k)"""
    lines = [l.rstrip("\n") for l in open(path) if not (l.startswith("!#=") or l.startswith("*")) and l.strip()]
    clusters = []
    i = 0
    while i < len(lines):
        head = lines[i]
        t = head[0]
        if t not in "GXY":
            raise ValueError("measurement type '%s' is not a GNSS type" % t)
        assert head[1] == " ", "ignored measurements are not expected in the fixture"
        k = 1 if t == "G" else int(head[42:62].split()[0])
        vscale = float(head[62:].split()[0])
        for s in head[62:].split()[1:4]:
            assert abs(float(s) - 1.0) < 1e-9, "p/l/h scalars are not expected in the fixture"
        if t == "Y":
            assert head[22:42].strip() == "XYZ"
        vectors = []
        V = np.zeros((3 * k, 3 * k))
        for j in range(k):
            h = lines[i]
            assert h[0] == t
            s1 = h[2:22].strip()
            s2 = "" if t == "Y" else h[22:42].strip()
            rows = [[float(v) for v in _NUM.findall(lines[i + 1 + r][62:])] for r in range(3)]
            assert [len(r) for r in rows] == [2, 3, 4], rows
            vectors.append((s1, s2, [rows[0][0], rows[1][0], rows[2][0]]))
            r0 = 3 * j
            for r in range(3):
                for c in range(r + 1):
                    V[r0 + c, r0 + r] = V[r0 + r, r0 + c] = rows[r][1 + c]
            i += 4
            for cb in range(k - 1 - j):
                c0 = 3 * (j + 1 + cb)
                for r in range(3):
                    vals = [float(v) for v in _NUM.findall(lines[i + r])]
                    assert len(vals) == 3, lines[i + r]
                    V[r0 + r, c0:c0 + 3] = vals
                    V[c0:c0 + 3, r0 + r] = vals
                i += 3
        clusters.append({"type": t, "vscale": vscale, "vectors": vectors, "V": V})
    return clusters


def read_adj(path):
    """the 'Adjusted Measurements' and 'Adjusted Coordinates' tables + summary figures of a simultaneous .adj report"""
    txt = open(path).read().split("\n")
    out = {"msr": [], "stn": {}}
    for l in txt:
        for key, tag in (("unknowns", "Number of unknown parameters"), ("measurements", "Number of measurements"),
                         ("dof", "Degrees of freedom"), ("chi2", "Chi squared"), ("sigma0", "Rigorous Sigma Zero")):
            if l.startswith(tag):
                out[key] = float(l[len(tag):].split()[0])
    i = next(n for n, l in enumerate(txt) if l.startswith("Adjusted Measurements"))
    i += 5
    while txt[i].strip():
        l = txt[i]
        f = l[67:].split()
        out["msr"].append({"type": l[0], "stn1": l[2:22].strip(), "stn2": l[22:42].strip(), "comp": l[65],
                           "measured": float(f[0]), "adjusted": float(f[1]), "correction": float(f[2]),
                           "meas_sd": float(f[3]), "adj_sd": float(f[4]), "corr_sd": float(f[5]), "nstat": float(f[6])})
        i += 1
    i = next(n for n, l in enumerate(txt) if l.startswith("Adjusted Coordinates"))
    i += 5
    while i < len(txt) and txt[i].strip():
        l = txt[i]
        f = l[26:].split()
        out["stn"][l[0:20].strip()] = {"xyz": [float(f[4]), float(f[5]), float(f[6])], "sd_enu": [float(f[7]), float(f[8]), float(f[9])]}
        i += 1
    return out


def write_binary_network(base, stations, clusters, measured=None):
    """.bst / .bms / .asl the way dnaimport lays GNSS measurements out (one record per X/Y/Z element, covariance
    records after each vector; dnaadjust.cpp:4214-4560 reads them back).  `measured` optionally replaces the
    observations (flat list, 3 per vector, file order)."""
    index = {s[0]: n for n, s in enumerate(stations)}
    bst = np.zeros(len(stations), dtype=F.STATION_DT)
    for n, (name, const, lat, lon, h, desc) in enumerate(stations):
        bst["stationName"][n] = name.encode()
        bst["stationNameOrig"][n] = name.encode()
        bst["stationConst"][n] = const.encode()
        bst["stationType"][n] = b"LLH"
        for k in ("initialLatitude", "currentLatitude"):
            bst[k][n] = lat
        for k in ("initialLongitude", "currentLongitude"):
            bst[k][n] = lon
        for k in ("initialHeight", "currentHeight"):
            bst[k][n] = h
        bst["description"][n] = desc.encode()[:128]
        bst["fileOrder"][n] = n
        bst["nameOrder"][n] = n
        bst["epsgCode"][n] = b"7843"
        bst["epoch"][n] = b"01.01.2020"
    recs = []
    counts = np.zeros(len(stations), dtype=np.uint32)
    q = 0
    for cid, cl in enumerate(clusters, start=1):
        t, k, V = cl["type"], len(cl["vectors"]), cl["V"]

        def rec(start, s1, s2, j):
            r = np.zeros(1, dtype=F.MEASUREMENT_DT)
            r["measType"] = t.encode()
            r["measStart"] = start
            r["measurementStations"] = 1 if t == "Y" else 2
            r["epsgCode"] = b"7843"
            r["epoch"] = b"01.01.2020"
            r["coordType"] = b"XYZ"
            r["station1"] = s1
            r["station2"] = s2
            r["vectorCount1"] = 1 if t == "G" else k
            r["vectorCount2"] = 0 if t == "G" else k - 1 - j
            r["clusterID"] = cid
            r["fileOrder"] = cid
            r["scale1"] = r["scale2"] = r["scale3"] = 1.0
            r["scale4"] = cl["vscale"]
            return r
        for j, (n1, n2, obs) in enumerate(cl["vectors"]):
            s1 = index[n1]
            s2 = 0 if t == "Y" else index[n2]
            counts[s1] += 1
            if t != "Y":
                counts[s2] += 1
            r0 = 3 * j
            for e in range(3):
                r = rec(e, s1, s2, j)
                r["term1"] = obs[e] if measured is None else measured[q]
                r["preAdjMeas"] = r["term1"]
                q += 1
                r["term2"] = V[r0, r0 + e]
                if e >= 1:
                    r["term3"] = V[r0 + 1, r0 + e]
                if e == 2:
                    r["term4"] = V[r0 + 2, r0 + 2]
                recs.append(r)
            for c in range(j + 1, k):
                for e in range(3):
                    r = rec(3 + e, s1, s2, j)
                    r["term1"], r["term2"], r["term3"] = V[r0 + e, 3 * c], V[r0 + e, 3 * c + 1], V[r0 + e, 3 * c + 2]
                    recs.append(r)
    bms = np.zeros(len(recs), dtype=F.MEASUREMENT_DT)     # (np.concatenate would repack the explicit-offset records)
    for n, r in enumerate(recs):
        bms[n] = r[0]
    F.write_bst(base + ".bst", bst)
    F.write_bms(base + ".bms", bms)
    F.write_asl(base + ".asl", counts)
    return bst, bms


def build_gnss_sample_with_the_product_importer(golden_dir, base):
    """the same network through the product's own importer (dnaimport_text, include/dnaadjust_c.h): .stn / .msr -> .bst / .bms / .asl
    with the ITRF2008 / ITRF2014 baselines aligned to GDA2020 the way dnareftran does -- nothing is lifted from the report.
    Returns (stations, clusters, expected) like build_gnss_sample."""
    import os
    from dynadjust_amd import adjust
    summary = adjust.import_dna_text(os.path.join(golden_dir, "gnss-network.stn"), os.path.join(golden_dir, "gnss-network.msr"), base)
    assert summary["stations"] == 43 and summary["vectors_transformed"] > 100
    stn = read_stn(os.path.join(golden_dir, "gnss-network.stn"))
    cl = read_msr(os.path.join(golden_dir, "gnss-network.msr"))
    adj = read_adj(os.path.join(golden_dir, "gnss.simult.adj.expected"))
    return stn, cl, adj


def build_gnss_sample(golden_dir, base):
    """the reference's sample GNSS network as .bst/.bms/.asl files at `base`.  The observations are the "Measured" column
    of gnss.simult.adj.expected rather than the .msr values: dnaimport transformed the ITRF2008/ITRF2014 baselines to the
    GDA2020 frame of the stations before the reference adjusted them (sub-millimetre to 1.5 mm changes), and that
    transformation (dnareftran) is out of scope.  Returns (stations, clusters, expected)."""
    import os
    stn = read_stn(os.path.join(golden_dir, "gnss-network.stn"))
    cl = read_msr(os.path.join(golden_dir, "gnss-network.msr"))
    adj = read_adj(os.path.join(golden_dir, "gnss.simult.adj.expected"))
    q = 0
    for c in cl:
        for (s1, s2, obs) in c["vectors"]:
            for e in range(3):
                m = adj["msr"][q]
                assert (m["type"], m["stn1"], m["stn2"], m["comp"]) == (c["type"], s1, s2, "XYZ"[e])
                assert abs(m["measured"] - obs[e]) < 0.005
                q += 1
    assert q == len(adj["msr"])
    write_binary_network(base, stn, cl, [m["measured"] for m in adj["msr"]])
    return stn, cl, adj
