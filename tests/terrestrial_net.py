"""Test-side generator of small mixed networks: terrestrial measurements (A B K C E M S V Z L H R I J P Q) over a station grid, a few
GNSS baselines and a GNSS point cluster for the datum, written as .bst/.bms/.asl/.seg with a strip segmentation.

The "truth" measurements are computed here with formulas written independently of the oracle and of the product
(numpy, local-frame vectors), so that a sign or frame slip on either side shows up as a network that does not come
back to its true coordinates."""
import math

import numpy as np

from . import dnaformats as F

A_GRS, INVF = 6378137.0, 298.257222101
E2 = 2 / INVF - 1 / INVF ** 2


def geo_to_cart(lat, lon, h):
    nu = A_GRS / math.sqrt(1 - E2 * math.sin(lat) ** 2)
    return np.array([(nu + h) * math.cos(lat) * math.cos(lon), (nu + h) * math.cos(lat) * math.sin(lon), (nu * (1 - E2) + h) * math.sin(lat)])


def cart_to_geo(X):
    x, y, z = X
    lon = math.atan2(y, x)
    p = math.hypot(x, y)
    lat = math.atan2(z, p * (1 - E2))
    for _ in range(15):
        nu = A_GRS / math.sqrt(1 - E2 * math.sin(lat) ** 2)
        h = p / math.cos(lat) - nu
        lat = math.atan2(z, p * (1 - E2 * nu / (nu + h)))
    return lat, lon, h


def enu_axes(lat, lon):
    e = np.array([-math.sin(lon), math.cos(lon), 0.0])
    n = np.array([-math.sin(lat) * math.cos(lon), -math.sin(lat) * math.sin(lon), math.cos(lat)])
    u = np.array([math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)])
    return e, n, u


class Builder:
    def __init__(self, rows, cols, blocks, seed=1, spacing=4.0e-5, defl=True, geoid=True, perturb=0.03):
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.rows, self.cols, self.blocks = rows, cols, blocks
        n = rows * cols
        self.n = n
        self.llh = np.zeros((n, 3))
        for r in range(rows):
            for c in range(cols):
                s = r * cols + c
                self.llh[s] = [math.radians(-36.5) + r * spacing, math.radians(146.0) + c * spacing * 1.2, 150.0 + 60.0 * rng.random()]
        self.truth = np.array([geo_to_cart(*p) for p in self.llh])
        self.geoid = (4.5 + 0.4 * rng.standard_normal(n)).astype(np.float32) if geoid else np.zeros(n, np.float32)
        sec = math.pi / 648000.0
        self.vdef = (6.0 * sec * rng.standard_normal(n)) if defl else np.zeros(n)      # prime vertical
        self.mdef = (5.0 * sec * rng.standard_normal(n)) if defl else np.zeros(n)      # meridian
        # initial coordinates: truth moved by a few centimetres
        self.init = self.truth + perturb * rng.standard_normal((n, 3))
        self.recs = []
        self.owner = []          # station that decides the block of each measurement
        self.counts = np.zeros(n, dtype=np.uint32)
        self.cid = 1

    # ---- independent measurement models at the truth ---------------------------------------------------------------
    def _local(self, s, t, ih=0.0, th=0.0):
        """e, n, up of the line (instrument at s, height ih) -> (target at t, height th) in the local frame of s"""
        es, ns_, us = enu_axes(self.llh[s][0], self.llh[s][1])
        _, _, ut = enu_axes(self.llh[t][0], self.llh[t][1])
        d = (self.truth[t] + th * ut) - (self.truth[s] + ih * us)
        return d @ es, d @ ns_, d @ us

    def azimuth(self, s, t):
        e, n, _ = self._local(s, t)
        return math.atan2(e, n) % (2 * math.pi)

    def zenith(self, s, t, ih, th):
        e, n, u = self._local(s, t, ih, th)
        return math.atan2(math.hypot(e, n), u)

    def slope(self, s, t, ih, th):
        # (the reference offsets BOTH heights along the instrument station's normal, dnaadjust.cpp:5466)
        _, _, us = enu_axes(self.llh[s][0], self.llh[s][1])
        return float(np.linalg.norm(self.truth[t] + th * us - (self.truth[s] + ih * us)))

    def chord(self, s, t):
        a = geo_to_cart(self.llh[s][0], self.llh[s][1], 0.0)
        b = geo_to_cart(self.llh[t][0], self.llh[t][1], 0.0)
        return float(np.linalg.norm(b - a))

    # ---- records -------------------------------------------------------------------------------------------------------
    def _rec(self, t, s1, s2=0, s3=0, value=0.0, var=1.0, ih=0.0, th=0.0, nstn=2):
        r = np.zeros(1, dtype=F.MEASUREMENT_DT)
        r["measType"] = t.encode()
        r["measStart"] = 0
        r["measurementStations"] = nstn
        r["epsgCode"] = b"7843"
        r["epoch"] = b"01.01.2020"
        r["coordType"] = b"XYZ"
        r["station1"], r["station2"], r["station3"] = s1, s2, s3
        r["vectorCount1"] = 0
        r["clusterID"] = self.cid
        r["fileOrder"] = self.cid
        self.cid += 1
        r["term1"], r["term2"], r["term3"], r["term4"] = value, var, ih, th
        r["scale1"] = r["scale2"] = r["scale3"] = r["scale4"] = 1.0
        r["preAdjMeas"] = value
        return r

    def add(self, t, s1, s2=None, s3=None, sd=None, ih=0.0, th=0.0):
        rng = self.rng
        sec = math.pi / 648000.0
        dV, dM = self.vdef[s1], self.mdef[s1]
        N1 = float(self.geoid[s1])
        if t == "S":
            sd = sd or 0.004
            v = self.slope(s1, s2, ih, th)
        elif t == "C":
            sd = sd or 0.004
            v = self.chord(s1, s2)
        elif t == "E":
            sd = sd or 0.004
            c = self.chord(s1, s2)
            az = self.azimuth(s1, s2)
            latm = 0.5 * (self.llh[s1][0] + self.llh[s2][0])
            nu = A_GRS / math.sqrt(1 - E2 * math.sin(latm) ** 2)
            rho = A_GRS * (1 - E2) / (1 - E2 * math.sin(latm) ** 2) ** 1.5
            R = rho * nu / (nu * math.cos(az) ** 2 + rho * math.sin(az) ** 2)
            v = 2 * R * math.asin(c / (2 * R))
        elif t == "M":
            sd = sd or 0.004
            c = self.chord(s1, s2)
            N2 = float(self.geoid[s2])
            latm = 0.5 * (self.llh[s1][0] + self.llh[s2][0])
            nu = A_GRS / math.sqrt(1 - E2 * math.sin(latm) ** 2)
            rho = A_GRS * (1 - E2) / (1 - E2 * math.sin(latm) ** 2) ** 1.5
            R = math.sqrt(nu * rho)
            mc = math.sqrt(c * c * (1 + N1 / R) * (1 + N2 / R) + (N2 - N1) ** 2)
            r_ = R + 0.5 * (N1 + N2)
            v = 2 * r_ * math.asin(mc / (2 * r_))
        elif t == "L":
            sd = sd or 0.002
            v = (self.llh[s2][2] - float(self.geoid[s2])) - (self.llh[s1][2] - N1)          # orthometric height difference
        elif t == "H":
            sd = sd or 0.01
            v = self.llh[s1][2] - N1
        elif t == "R":
            sd = sd or 0.01
            v = self.llh[s1][2]
        elif t == "B":
            sd = sd or 2.0 * sec
            v = self.azimuth(s1, s2)
        elif t == "K":
            # astronomic azimuth = geodetic + Laplace correction
            sd = sd or 2.0 * sec
            az = self.azimuth(s1, s2)
            z = self.zenith(s1, s2, ih, th)
            v = az + dV * math.tan(self.llh[s1][0]) + (dM * math.sin(az) - dV * math.cos(az)) / math.tan(z)
        elif t == "V":
            # observed zenith distance refers to the plumb line: geodetic minus the deflection component along the line
            sd = sd or 3.0 * sec
            az = self.azimuth(s1, s2)
            v = self.zenith(s1, s2, ih, th) - (dM * math.cos(az) + dV * math.sin(az))
        elif t == "Z":
            sd = sd or 3.0 * sec
            az = self.azimuth(s1, s2)
            v = (math.pi / 2 - self.zenith(s1, s2, ih, th)) + (dM * math.cos(az) + dV * math.sin(az))
        elif t == "A":
            sd = sd or 2.0 * sec
            a12, a13 = self.azimuth(s1, s2), self.azimuth(s1, s3)
            z12, z13 = self.zenith(s1, s2, ih, th), self.zenith(s1, s3, ih, th)
            c12 = (dM * math.sin(a12) - dV * math.cos(a12)) / math.tan(z12)
            c13 = (dM * math.sin(a13) - dV * math.cos(a13)) / math.tan(z13)
            v = ((a13 - a12) % (2 * math.pi)) + (c13 - c12)
        elif t == "P":
            sd = sd or 0.0004 * sec                 # ~ 12 mm on the ground
            v = self.llh[s1][0]
        elif t == "Q":
            sd = sd or 0.0005 * sec
            v = self.llh[s1][1]
        elif t == "I":
            # astronomic latitude = geodetic + meridian component of the deflection
            sd = sd or 0.0004 * sec
            v = self.llh[s1][0] + dM
        elif t == "J":
            # astronomic longitude = geodetic + prime-vertical component x sec(latitude)
            sd = sd or 0.0005 * sec
            v = self.llh[s1][1] + dV / math.cos(self.llh[s1][0])
        else:
            raise ValueError(t)
        v += sd * rng.standard_normal()
        nstn = 3 if t == "A" else (1 if t in "HRIJPQ" else 2)
        self.recs.append(self._rec(t, s1, s2 or 0, s3 or 0, v, sd * sd, ih, th, nstn))
        self.owner.append(min(x for x in (s1, s2, s3) if x is not None))
        for x in (s1, s2, s3):
            if x is not None:
                self.counts[x] += 1

    def add_directions(self, inst, targets, sd=None):
        """a direction set: one round of horizontal directions from `inst` with an unknown orientation of the circle.  What the
        theodolite reads refers to the plumb line: geodetic azimuth + (xi sin az - eta cos az) cot z, plus the orientation"""
        rng = self.rng
        sec = math.pi / 648000.0
        sd = sd or 2.0 * sec
        dV, dM = self.vdef[inst], self.mdef[inst]
        omega = rng.uniform(0.0, 2 * math.pi)
        cid = self.cid
        self.cid += 1
        # now and then one direction of a larger round is flagged "ignore" (a blunder far off its true value): the set then
        # skips it (dnaadjust.cpp:5119-5130) and vectorCount2 counts the others
        dropped = int(rng.integers(1, len(targets))) if (len(targets) >= 4 and rng.random() < 0.5) else -1
        for j, t in enumerate(targets):
            az = self.azimuth(inst, t)
            z = self.zenith(inst, t, 0.0, 0.0)
            v = (az + (dM * math.sin(az) - dV * math.cos(az)) / math.tan(z) + omega + sd * rng.standard_normal()) % (2 * math.pi)
            if j == dropped:
                v = (v + 0.3) % (2 * math.pi)
            r = self._rec("D", inst, t, 0, v, sd * sd, 0.0, 0.0, 2)
            r["clusterID"] = cid
            if j == 0:
                r["measStart"] = 0
                r["vectorCount1"] = len(targets)                            # directions of the set, the first included
                r["vectorCount2"] = len(targets) - (1 if dropped > 0 else 0)   # ... of which not ignored
            else:
                r["measStart"] = 1
                r["ignore"] = j == dropped
            self.recs.append(r)
            self.owner.append(min([inst] + list(targets)))
            self.counts[t] += 1
        self.counts[inst] += 1
        self.cid = cid + 1

    def add_baseline(self, s1, s2, sd=0.003):
        d = self.truth[s2] - self.truth[s1] + sd * self.rng.standard_normal(3)
        V = np.eye(3) * sd * sd
        cid = self.cid
        self.cid += 1
        first = len(self.recs)
        for e in range(3):
            r = self._rec("G", s1, s2, 0, d[e], 0.0)
            r["measStart"] = e
            r["clusterID"] = cid
            r["vectorCount1"] = 1 if e == 0 else 0
            r["term2"] = V[0, e]
            r["term3"] = V[1, e] if e >= 1 else 0.0
            r["term4"] = V[2, 2] if e == 2 else 0.0
            self.recs.append(r)
        self.cid = cid + 1
        self.owner += [min(s1, s2)] * 3
        self.counts[s1] += 1
        self.counts[s2] += 1
        return first

    def add_point(self, s, sd=0.002):
        d = self.truth[s] + sd * self.rng.standard_normal(3)
        cid = self.cid
        for e in range(3):
            r = self._rec("Y", s, 0, 0, d[e], 0.0, nstn=1)
            r["measStart"] = e
            r["clusterID"] = cid
            r["vectorCount1"] = 1
            r["vectorCount2"] = 0
            r["term2"] = sd * sd if e == 0 else 0.0
            r["term3"] = sd * sd if e == 1 else 0.0
            r["term4"] = sd * sd if e == 2 else 0.0
            self.recs.append(r)
        self.cid = cid + 1
        self.owner += [s] * 3
        self.counts[s] += 1

    # ---- files -----------------------------------------------------------------------------------------------------------
    def write(self, base):
        n = self.n
        bst = np.zeros(n, dtype=F.STATION_DT)
        for s in range(n):
            lat, lon, h = cart_to_geo(self.init[s])
            bst["stationName"][s] = ("T%05d" % s).encode()
            bst["stationNameOrig"][s] = bst["stationName"][s]
            bst["stationConst"][s] = b"FFF"
            bst["stationType"][s] = b"LLH"
            for k in ("initialLatitude", "currentLatitude"):
                bst[k][s] = lat
            for k in ("initialLongitude", "currentLongitude"):
                bst[k][s] = lon
            for k in ("initialHeight", "currentHeight"):
                bst[k][s] = h
            bst["geoidSep"][s] = self.geoid[s]
            bst["verticalDef"][s] = self.vdef[s]
            bst["meridianDef"][s] = self.mdef[s]
            bst["fileOrder"][s] = bst["nameOrder"][s] = s
            bst["epsgCode"][s] = b"7843"
            bst["epoch"][s] = b"01.01.2020"
        bms = np.zeros(len(self.recs), dtype=F.MEASUREMENT_DT)
        for i, r in enumerate(self.recs):
            bms[i] = r[0]
        F.write_bst(base + ".bst", bst)
        F.write_bms(base + ".bms", bms)
        F.write_asl(base + ".asl", self.counts)
        # strip segmentation: rows split into `blocks` strips; a measurement belongs to the strip of its lowest station;
        # stations of a later strip it touches are junctions of that block
        R, B, Cc = self.rows, self.blocks, self.cols
        strip = lambda s: (s // Cc) * B // R
        ISL = [[] for _ in range(B)]
        JSL = [[] for _ in range(B)]
        CML = [[] for _ in range(B)]
        for s in range(n):
            ISL[strip(s)].append(s)
        for i, r in enumerate(bms):
            if r["measStart"] != 0:
                continue
            k = strip(self.owner[i])
            CML[k].append(i)
            stns = [int(r["station1"])]
            if r["measType"] == b"D":
                stns += [int(bms["station2"][i + j]) for j in range(int(r["vectorCount1"]))]
            elif r["measurementStations"] >= 2 and r["measType"] != b"Y":
                stns.append(int(r["station2"]))
            if r["measurementStations"] >= 3:
                stns.append(int(r["station3"]))
            for s in stns:
                assert strip(s) in (k, k + 1), "measurement spans more than two strips"
                if strip(s) != k and s not in JSL[k]:
                    JSL[k].append(s)
        F.write_seg(base + ".seg", ISL, [sorted(j) for j in JSL], CML, [0] * B, bms)
        np.asarray(self.truth).ravel().tofile(base + ".truth")
        return bst, bms


def build_oscillating_network(base, blocks=2, short=0.05, perturb=0.8, seed=4):
    """A network whose iteration cannot settle: two stations (the middle of the first and of the third row) are each held by nothing but
    two slope distances to their row neighbours -- nearly in line with them -- that are both `short` metres too short to meet, and an
    ellipsoidal height.  Across the line the Gauss-Newton step then overshoots the line from either side (y -> y/2 - d short / y for a
    line of half-length d): corrections of metres that turn round from iteration to iteration.  Every other station carries a GNSS
    point.  4 x 3 stations, `blocks` strips; slope distances down the outer columns tie the strips together."""
    b = Builder(4, 3, blocks, seed, defl=False, geoid=False, perturb=0.02)
    rng = b.rng
    loose = (1, 7)
    for s in (0, 1, 2, 6, 7, 8):            # the two rows level: the loose stations in line with their neighbours (but for the parallel's sagitta)
        b.llh[s][2] = 150.0
        b.truth[s] = geo_to_cart(*b.llh[s])
        b.init[s] = b.truth[s] + 0.02 * rng.standard_normal(3)
    for s in range(b.n):
        if s not in loose:
            b.add_point(s)
    for s in loose:
        e, n, u = enu_axes(b.llh[s][0], b.llh[s][1])
        b.init[s] = b.truth[s] + perturb * n + 0.01 * rng.standard_normal(3)       # off the line, across it
        for t in (s - 1, s + 1):
            b.add("S", min(s, t), max(s, t), ih=0.0, th=0.0)
            b.recs[-1]["term1"] -= short
            b.recs[-1]["preAdjMeas"] = b.recs[-1]["term1"]
        b.add("R", s)
    for c in (0, 2):                        # (not the middle column: a distance along it would hold the loose stations across their lines)
        for r in range(3):
            b.add("S", r * 3 + c, (r + 1) * 3 + c)
    return b, b.write(base)


def build_mixed_network(base, rows=6, cols=5, blocks=1, seed=1, types="SVZLHRBKACEM", defl=True, geoid=True):
    """a grid network observed with every terrestrial type in `types`, plus GNSS baselines along the first column and one
    GNSS point per corner for the datum"""
    b = Builder(rows, cols, blocks, seed, defl=defl, geoid=geoid)
    rng = b.rng
    ih = lambda: 1.4 + 0.3 * rng.random()
    for r in range(rows):
        for c in range(cols):
            s = r * cols + c
            nb = []
            if c + 1 < cols:
                nb.append(s + 1)
            if r + 1 < rows:
                nb.append(s + cols)
            if r + 1 < rows and c + 1 < cols:
                nb.append(s + cols + 1)
            for t in nb:
                for ty in types:
                    if ty in "SVZ":
                        b.add(ty, s, t, ih=ih(), th=ih())
                    elif ty == "L":
                        b.add(ty, s, t)              # levelling along every line: the heights are what the other types barely see
                    elif ty in "CEMBK":
                        if rng.random() < 0.5:
                            b.add(ty, s, t, ih=ih(), th=ih()) if ty == "K" else b.add(ty, s, t)
            if "A" in types and len(nb) >= 2:
                b.add("A", s, nb[0], nb[1], ih=ih(), th=ih())
                if len(nb) == 3:
                    b.add("A", s, nb[2], nb[0], ih=ih(), th=ih())
            for ty in "HRIJPQ":
                if ty in types and rng.random() < 0.4:
                    b.add(ty, s)
            if "D" in types and len(nb) >= 2:
                back = [x for x in (s - 1, s - cols) if x >= 0 and (x // cols == s // cols or x == s - cols) and (s // cols) - (x // cols) <= 0]
                tg = nb + [x for x in back if x // cols == s // cols]      # neighbours of this and the next row only: two strips at most
                rng.shuffle(tg)
                b.add_directions(s, tg)
    for r in range(rows - 1):
        b.add_baseline(r * cols, (r + 1) * cols)
    for s in (0, cols - 1, (rows - 1) * cols, rows * cols - 1):
        b.add_point(s)
    return b, b.write(base)
