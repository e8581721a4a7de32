"""Test-only stand-in for `dnaimport` + `dnageoid` + `dnasegment` on the reference's urban sample network
(sampleData/urban-network.{stn,msr,geo} with urban.phased.adj.expected, copied as DATA under tests/golden/): turns the
text files into .bst/.bms/.asl/.seg and reads the expected tables of the report.  The product never reads these formats
(import, geoid interpolation and segmentation are out of scope).

What the pipeline of the reference's own test does (CMakeLists.txt:1062-1070): import (UTM stations -> geographic, angles to
radians, standard deviations to variances), geoid (N and deflections per station from an NTv2 grid; orthometric station
heights -> ellipsoidal), segment (2 blocks), adjust --phased.  Here: the same conversions, N and the deflections from the
exported urban-network.geo (3 decimals; N refined to 4 decimals from the h - H columns of the expected report), and a
two-block segmentation of our own -- a phased adjustment is rigorous, its results do not depend on where the cut is."""
import math
import os

import numpy as np

from . import dnaformats as F
from . import dnatext as T

SEC = math.pi / 648000.0
ANGULAR = "ABKVZ"
HEIGHTS = "SVZ"            # types with instrument / target heights in the file
XYZ_T, LLh_T, LLH_T, UTM_T = 0, 1, 2, 3


def utm_to_geo(easting, northing, zone, a=6378137.0, invf=298.257222101, k0=0.9996, fe=500000.0, fn=10000000.0):
    """MGA / UTM grid -> latitude, longitude (radians), Krueger series (GDA technical manual), southern hemisphere"""
    f = 1.0 / invf
    n = f / (2.0 - f)
    A = a / (1.0 + n) * (1.0 + n ** 2 / 4.0 + n ** 4 / 64.0 + n ** 6 / 256.0)
    beta = [n / 2 - 2 * n ** 2 / 3 + 37 * n ** 3 / 96 - n ** 4 / 360, n ** 2 / 48 + n ** 3 / 15 - 437 * n ** 4 / 1440,
            17 * n ** 3 / 480 - 37 * n ** 4 / 840, 4397 * n ** 4 / 161280]
    delta = [2 * n - 2 * n ** 2 / 3 - 2 * n ** 3 + 116 * n ** 4 / 45, 7 * n ** 2 / 3 - 8 * n ** 3 / 5 - 227 * n ** 4 / 45,
             56 * n ** 3 / 15 - 136 * n ** 4 / 35, 4279 * n ** 4 / 630]
    xi = (northing - fn) / (k0 * A)
    eta = (easting - fe) / (k0 * A)
    xi1 = xi - sum(b * math.sin(2 * (j + 1) * xi) * math.cosh(2 * (j + 1) * eta) for j, b in enumerate(beta))
    eta1 = eta - sum(b * math.cos(2 * (j + 1) * xi) * math.sinh(2 * (j + 1) * eta) for j, b in enumerate(beta))
    chi = math.asin(math.sin(xi1) / math.cosh(eta1))
    lat = chi + sum(d * math.sin(2 * (j + 1) * chi) for j, d in enumerate(delta))
    lon = math.radians(zone * 6 - 183) + math.atan2(math.sinh(eta1), math.cos(xi1))
    return lat, lon


def read_stations(path):
    """-> list of dict(name, const, type, lat, lon, height): UTM (orthometric heights in this file) or LLH records"""
    out = []
    for line in open(path):
        if line.startswith("!#=") or line.startswith("*") or not line.strip():
            continue
        name, const, ctype = line[0:20].strip(), line[20:23], line[24:27]
        v = line[27:].split()
        if ctype == "UTM":
            lat, lon = utm_to_geo(float(v[0]), float(v[1]), int(v[3]))
            out.append(dict(name=name, const=const, type=UTM_T, lat=lat, lon=lon, height=float(v[2])))
        elif ctype == "LLH":
            out.append(dict(name=name, const=const, type=LLH_T, lat=math.radians(T.dms_to_deg(float(v[0]))),
                            lon=math.radians(T.dms_to_deg(float(v[1]))), height=float(v[2])))
        else:
            raise ValueError("station coordinate type " + ctype)
    return out


def read_geo(path):
    """DNA geoid file (dnageoid.cpp:815): name -> (N [m], deflection in the meridian [rad], in the prime vertical [rad])"""
    out = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("!") or line.startswith("*") or not line.strip():
            continue
        name = line[0:41].strip()
        n, dm, dv = (float(x) for x in line[41:].split()[:3])
        out[name] = (n, dm * SEC, dv * SEC)
    return out


def _dms(tokens):
    """['-0', '37', '11.0000'] -> radians"""
    sign = -1.0 if tokens[0].startswith("-") else 1.0
    d, m, s = abs(float(tokens[0])), float(tokens[1]), float(tokens[2])
    return sign * math.radians(d + m / 60.0 + s / 3600.0)


def read_measurements(path):
    """file order list of dicts.  Terrestrial: type, ignore, stn (1-3 names), value (rad / m), var, ih, th.
    GNSS: type, ignore, coord ('XYZ' / 'LLH' ...), vscale, vectors [(stn1, stn2, [3 values])], V (3k x 3k as in the file)"""
    lines = [l.rstrip("\n") for l in open(path) if not (l.startswith("!#=") or l.startswith("*")) and l.strip()]
    out = []
    i = 0
    while i < len(lines):
        h = lines[i]
        t, ignore = h[0], h[1] == "*"
        if t in "GXY":
            k = 1 if t == "G" else int(h[42:62].split()[0])
            coord = h[22:42].strip() if t == "Y" else "XYZ"
            sc = h[62:].split()
            vscale = float(sc[0])
            assert all(abs(float(x) - 1.0) < 1e-9 for x in sc[:4]), "scalars are not expected in this fixture"
            vectors, V = [], np.zeros((3 * k, 3 * k))
            for j in range(k):
                hh = lines[i]
                s1 = hh[2:22].strip()
                s2 = "" if t == "Y" else hh[22:42].strip()
                rows = [T._NUM.findall(lines[i + 1 + r][62:]) for r in range(3)]
                assert [len(r) for r in rows] == [2, 3, 4], rows
                vals = [float(rows[r][0]) for r in range(3)]
                if t == "Y" and coord in ("LLH", "LLh"):
                    vals[0], vals[1] = math.radians(T.dms_to_deg(vals[0])), math.radians(T.dms_to_deg(vals[1]))
                vectors.append((s1, s2, vals))
                r0 = 3 * j
                for r in range(3):
                    for c in range(r + 1):
                        V[r0 + c, r0 + r] = V[r0 + r, r0 + c] = float(rows[r][1 + c])
                i += 4
                for cb in range(k - 1 - j):
                    c0 = 3 * (j + 1 + cb)
                    for r in range(3):
                        v3 = [float(x) for x in T._NUM.findall(lines[i + r])]
                        assert len(v3) == 3, lines[i + r]
                        V[r0 + r, c0:c0 + 3] = v3
                        V[c0:c0 + 3, r0 + r] = v3
                    i += 3
            out.append(dict(type=t, ignore=ignore, coord=coord, vscale=vscale, vectors=vectors, V=V))
            continue
        stn = [h[2:22].strip(), h[22:42].strip(), h[42:62].strip()]
        stn = [s for s in stn if s]
        tok = h[62:].split()
        if t in ANGULAR:
            value, sd, rest = _dms(tok[0:3]), float(tok[3]) * SEC, tok[4:]
        else:
            value, sd, rest = float(tok[0]), float(tok[1]), tok[2:]
        ih, th = (float(rest[0]), float(rest[1])) if (t in HEIGHTS and len(rest) >= 2) else (0.0, 0.0)
        assert t in "ABCEHKLMRSVZ", t
        out.append(dict(type=t, ignore=ignore, stn=stn, value=value, var=sd * sd, ih=ih, th=th))
        i += 1
    return out


def read_report(path):
    """urban.phased.adj.expected: summary figures, the adjusted-measurement table (angles in radians, their corrections and
    standard deviations in radians too) and the adjusted-coordinate table"""
    txt = open(path).read().split("\n")
    out = {"msr": [], "stn": {}}
    for l in txt:
        for key, tag in (("unknowns", "Number of unknown parameters"), ("measurements", "Number of measurements"),
                         ("dof", "Degrees of freedom"), ("chi2", "Chi squared"), ("sigma0", "Rigorous Sigma Zero"),
                         ("pelzer", "Global (Pelzer) Reliability")):
            if l.startswith(tag):
                out[key] = float(l[len(tag):].split()[0])
        if l.startswith("Number of measurements") and "potential outliers" in l:
            out["outliers"] = int(l.split("(")[1].split()[0])
    i = next(n for n, l in enumerate(txt) if l.startswith("Adjusted Measurements")) + 5
    while txt[i].strip():
        l = txt[i]
        t, comp = l[0], l[65]
        f = l[67:].split()
        angular = t in ANGULAR or (t == "Y" and comp in "PL")
        if angular:
            measured, adjusted, f = _dms(f[0:3]), _dms(f[3:6]), f[6:]
            u = SEC
        else:
            measured, adjusted, f = float(f[0]), float(f[1]), f[2:]
            u = 1.0
        out["msr"].append(dict(type=t, stn=[s for s in (l[2:22].strip(), l[22:42].strip(), l[42:62].strip()) if s], comp=comp.strip(),
                               measured=measured, adjusted=adjusted, correction=float(f[0]) * u, meas_sd=float(f[1]) * u,
                               adj_sd=float(f[2]) * u, corr_sd=float(f[3]) * u, nstat=float(f[4]), pelzer=float(f[5]),
                               pre_adj_corr=float(f[6]) * u, unit=u))
        i += 1
    i = next(n for n, l in enumerate(txt) if l.startswith("Adjusted Coordinates")) + 5
    while i < len(txt) and txt[i].strip():
        l = txt[i]
        f = l[20:].split()
        out["stn"][l[0:20].strip()] = dict(const=f[0], H=float(f[3]), h=float(f[4]), xyz=[float(f[5]), float(f[6]), float(f[7])],
                                           sd_enu=[float(f[8]), float(f[9]), float(f[10])])
        i += 1
    return out


SAMPLES = {
    # the reference's test 2 (CMakeLists.txt:1062-1070): GDA94 as supplied, sequential phased adjustment, 2 blocks
    "gda94": ("urban-network.stn", "urban-network.msr", "urban.phased.adj.expected"),
    # its test 3 (CMakeLists.txt:1076-1083): the same network after dnareftran -r gda2020 (stations and GNSS measurements moved to
    # GDA2020: the transformed files the reference itself exported), multi-thread phased adjustment, 3 blocks
    "gda2020": ("urban.GDA2020.1.1.2020.stn", "urban.GDA2020.1.1.2020.msr", "urban_mt.phased-mt.adj.expected"),
}


def build_urban_sample(golden_dir, base, blocks=2, sample="gda94"):
    """the sample as .bst/.bms/.asl/.seg at `base`; returns (stations, measurements, report, bst, bms, cml_of_record)"""
    f_stn, f_msr, f_rep = SAMPLES[sample]
    stations = read_stations(os.path.join(golden_dir, f_stn))
    if sample == "gda2020":
        # The exported GDA2020 station file (reftran 1.0.3, 2020) also moved the orthometric heights by the change of the
        # ellipsoidal height between the frames (-0.089 m); the build that produced the expected report (1.2.9, 2025) left them
        # as supplied -- its H(Ortho) column equals the GDA94 run's.  Constrained stations are held at their supplied heights,
        # so the heights come from the supplied file, the horizontal position from the transformed one.
        supplied = {s["name"]: s["height"] for s in read_stations(os.path.join(golden_dir, "urban-network.stn"))}
        for s in stations:
            s["height"] = supplied[s["name"]]
    msrs = read_measurements(os.path.join(golden_dir, f_msr))
    geo = read_geo(os.path.join(golden_dir, "urban-network.geo"))
    rep = read_report(os.path.join(golden_dir, f_rep))
    index = {s["name"]: n for n, s in enumerate(stations)}
    n = len(stations)
    bst = np.zeros(n, dtype=F.STATION_DT)
    for k, s in enumerate(stations):
        N, dm, dv = geo[s["name"]]
        if s["name"] in rep["stn"]:
            # the report prints H and h to 4 decimals: N to 1e-4 instead of the .geo file's 1e-3
            r = rep["stn"][s["name"]]
            if abs((r["h"] - r["H"]) - N) < 1.0e-3:
                N = r["h"] - r["H"]
        bst["stationName"][k] = bst["stationNameOrig"][k] = s["name"].encode()
        bst["stationConst"][k] = s["const"].encode()
        bst["stationType"][k] = b"UTM" if s["type"] == UTM_T else b"LLH"
        bst["suppliedStationType"][k] = s["type"]
        for key in ("initialLatitude", "currentLatitude"):
            bst[key][k] = s["lat"]
        for key in ("initialLongitude", "currentLongitude"):
            bst[key][k] = s["lon"]
        for key in ("initialHeight", "currentHeight"):
            bst[key][k] = s["height"] + N             # dnageoid: orthometric -> ellipsoidal
        bst["geoidSep"][k] = N
        bst["meridianDef"][k] = dm
        bst["verticalDef"][k] = dv
        bst["fileOrder"][k] = bst["nameOrder"][k] = k
        bst["epsgCode"][k] = b"4283"
        bst["epoch"][k] = b"01.01.1994"
    recs, first_of, counts = [], [], np.zeros(n, dtype=np.uint32)

    def rec(t, start, cid, ignore):
        r = np.zeros(1, dtype=F.MEASUREMENT_DT)
        r["measType"] = t.encode()
        r["measStart"] = start
        r["ignore"] = ignore
        r["epsgCode"] = b"4283"
        r["epoch"] = b"01.01.1994"
        r["coordType"] = b"XYZ"
        r["clusterID"] = r["fileOrder"] = cid
        r["scale1"] = r["scale2"] = r["scale3"] = r["scale4"] = 1.0
        return r
    for cid, m in enumerate(msrs, start=1):
        t = m["type"]
        first_of.append(len(recs))
        if t in "GXY":
            k, V = len(m["vectors"]), m["V"]
            for j, (n1, n2, obs) in enumerate(m["vectors"]):
                s1 = index[n1]
                s2 = 0 if t == "Y" else index[n2]
                if not m["ignore"]:
                    counts[s1] += 1
                    if t != "Y":
                        counts[s2] += 1
                r0 = 3 * j
                for e in range(3):
                    r = rec(t, e, cid, m["ignore"])
                    r["measurementStations"] = 1 if t == "Y" else 2
                    r["coordType"] = m["coord"].encode()
                    r["station1"], r["station2"] = s1, s2
                    r["vectorCount1"] = 1 if t == "G" else k
                    r["vectorCount2"] = 0 if t == "G" else k - 1 - j
                    r["scale4"] = m["vscale"]
                    r["term1"] = r["preAdjMeas"] = obs[e]
                    r["term2"] = V[r0, r0 + e]
                    if e >= 1:
                        r["term3"] = V[r0 + 1, r0 + e]
                    if e == 2:
                        r["term4"] = V[r0 + 2, r0 + 2]
                    recs.append(r)
                for c in range(j + 1, k):
                    for e in range(3):
                        r = rec(t, 3 + e, cid, m["ignore"])
                        r["measurementStations"] = 1 if t == "Y" else 2
                        r["coordType"] = m["coord"].encode()
                        r["station1"], r["station2"] = s1, s2
                        r["vectorCount1"], r["vectorCount2"] = k, k - 1 - j
                        r["term1"], r["term2"], r["term3"] = V[r0 + e, 3 * c], V[r0 + e, 3 * c + 1], V[r0 + e, 3 * c + 2]
                        recs.append(r)
            continue
        ids = [index[s] for s in m["stn"]]
        r = rec(t, 0, cid, m["ignore"])
        r["measurementStations"] = len(ids)
        r["station1"] = ids[0]
        r["station2"] = ids[1] if len(ids) > 1 else 0
        r["station3"] = ids[2] if len(ids) > 2 else 0
        r["term1"] = r["preAdjMeas"] = m["value"]
        r["term2"], r["term3"], r["term4"] = m["var"], m["ih"], m["th"]
        if not m["ignore"]:
            for s in ids:
                counts[s] += 1
        recs.append(r)
    bms = np.zeros(len(recs), dtype=F.MEASUREMENT_DT)
    for k, r in enumerate(recs):
        bms[k] = r[0]
    F.write_bst(base + ".bst", bst)
    F.write_bms(base + ".bms", bms)
    F.write_asl(base + ".asl", counts)
    write_cut_segmentation(base + ".seg", bst, bms, [first_of[c] for c, m in enumerate(msrs) if not m["ignore"]], blocks)
    return stations, msrs, rep, bst, bms, first_of


def write_cut_segmentation(path, bst, bms, cml, blocks):
    """`blocks` blocks by easting order: block k owns (as inner stations) the k-th slice of the stations sorted by longitude
    that no earlier block holds; a measurement goes to the first block that owns one of its stations; the other stations of
    that block's measurements are its junction stations and become inner (or stay junction) further on -- the invariant the
    reference's segmentation guarantees (JSL(k) inside the stations of block k+1, dnasegment.cpp:529)."""
    n = len(bst)

    def stations_of(i):
        r = bms[i]
        s = [int(r["station1"])]
        if r["measType"] not in (b"Y", b"H", b"R", b"I", b"J", b"P", b"Q"):
            s.append(int(r["station2"]))
        if r["measType"] == b"A":
            s.append(int(r["station3"]))
        if r["measType"] in (b"X", b"Y"):          # every vector of the cluster
            k, j, q = int(r["vectorCount1"]), 0, i
            s = []
            while j < k:
                rr = bms[q]
                s.append(int(rr["station1"]))
                if r["measType"] == b"X":
                    s.append(int(rr["station2"]))
                q += 3 + 3 * int(rr["vectorCount2"])
                j += 1
        return sorted(set(s))
    order = np.argsort(bst["currentLongitude"], kind="stable")
    used = [s for i in cml for s in stations_of(i)]
    active = sorted(set(used))
    slice_of = {}
    for rank, s in enumerate([int(s) for s in order if int(s) in set(active)]):
        slice_of[s] = min(blocks - 1, rank * blocks // len(active))
    msr_stn = {i: stations_of(i) for i in cml}
    ISL, JSL, CML = [[] for _ in range(blocks)], [[] for _ in range(blocks)], [[] for _ in range(blocks)]
    for i in cml:
        CML[min(slice_of[s] for s in msr_stn[i])].append(i)
    for s in active:
        ISL[slice_of[s]].append(s)
    for k in range(blocks):
        inner = set(ISL[k])
        carried = set(JSL[k - 1]) if k else set()
        # junctions carried in that this block's own measurements do not finish off stay junction... keep it simple: a carried
        # station whose slice is later than k stays junction, one of slice k is inner here (already in ISL[k])
        touched = set(s for i in CML[k] for s in msr_stn[i]) | carried
        JSL[k] = sorted(s for s in touched if s not in inner and slice_of[s] > k)
        assert all(slice_of[s] >= k for s in touched), "a measurement reaches back over a finished block"
    assert not JSL[-1]
    F.write_seg(path, [sorted(x) for x in ISL], JSL, [sorted(x) for x in CML], [0] * blocks, bms)
