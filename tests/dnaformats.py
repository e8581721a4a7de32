"""Independent (numpy) readers of DynAdjust's .bst / .bms / .asl / .seg files, used only by
the tests to hand the same network to the CPU oracle that the product reads through its own
C++ readers.  Byte layouts: SURVEY.md 8(b) (station_t 352 B, measurement_t 208 B, 60-byte
file info + metadata, include/io/dynadjust_file.cpp:83-181 of the reference)."""
import struct

import numpy as np

STATION_DT = np.dtype({
    "names": ["stationName", "stationNameOrig", "stationConst", "stationType", "suppliedStationType",
              "initialLatitude", "currentLatitude", "initialLongitude", "currentLongitude", "initialHeight", "currentHeight",
              "suppliedHeightRefFrame", "geoidSep", "geoidSepUnc", "meridianDef", "verticalDef", "zone", "description",
              "fileOrder", "nameOrder", "clusterID", "unusedStation", "epsgCode", "epoch", "observation_epoch", "plate"],
    "formats": ["S31", "S40", "S4", "S4", "<u2", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8", "<u2", "<f4", "<f4", "<f8", "<f8", "<i2",
                "S129", "<u4", "<u4", "<u4", "<u2", "S7", "S12", "S12", "S3"],
    "offsets": [0, 31, 71, 75, 80, 88, 96, 104, 112, 120, 128, 136, 140, 144, 152, 160, 168, 170, 300, 304, 308, 312, 314, 321, 333, 345],
    "itemsize": 352})

MEASUREMENT_DT = np.dtype({
    "names": ["measType", "measStart", "measurementStations", "epsgCode", "epoch", "observation_epoch", "coordType", "ignore",
              "station1", "station2", "station3", "vectorCount1", "vectorCount2", "clusterID", "fileOrder", "sourceFileIndex",
              "term1", "term2", "term3", "term4", "scale1", "scale2", "scale3", "scale4",
              "measAdj", "measCorr", "measAdjPrec", "residualPrec", "NStat", "TStat", "PelzerRel", "preAdjCorr", "preAdjMeas"],
    "formats": ["S1", "i1", "i1", "S7", "S12", "S12", "S4", "?", "<u4", "<u4", "<u4", "<u4", "<u4", "<u4", "<u4", "<u4"] + ["<f8"] * 17,
    "offsets": [0, 1, 2, 3, 10, 22, 34, 38, 40, 44, 48, 52, 56, 60, 64, 68] + [72 + 8 * i for i in range(17)],
    "itemsize": 208})


def _read_header(f):
    info = f.read(60)
    assert info[:10] == b"VERSION   " and info[20:30] == b"CREATED ON" and info[40:50] == b"CREATED BY", info
    return info[10:20].decode().strip()


def _read_meta(f, version):
    major, minor = (int(x) for x in version.split("."))
    v12 = (major, minor) >= (1, 2)
    (count,) = struct.unpack("<Q", f.read(8))
    reduced = f.read(1) != b"\x00"
    f.read(20)  # modifiedBy
    f.read(7)   # epsg
    f.read(12)  # epoch
    if v12:
        f.read(12)
    f.read(2)   # reftran, geoid
    (nin,) = struct.unpack("<Q", f.read(8))
    for _ in range(nin):
        f.read(256 + 7 + 12 + (12 if v12 else 0) + 2 + 2)
    if (major, minor) >= (1, 1):
        (nsrc,) = struct.unpack("<Q", f.read(8))
        f.read(256 * nsrc)
    return count, reduced


def read_bst(path):
    with open(path, "rb") as f:
        ver = _read_header(f)
        count, _ = _read_meta(f, ver)
        return np.frombuffer(f.read(count * 352), dtype=STATION_DT, count=count)


def read_bms(path):
    with open(path, "rb") as f:
        ver = _read_header(f)
        count, _ = _read_meta(f, ver)
        return np.frombuffer(f.read(count * 208), dtype=MEASUREMENT_DT, count=count)


def read_asl(path):
    with open(path, "rb") as f:
        _read_header(f)
        (count,) = struct.unpack("<Q", f.read(8))
        raw = np.frombuffer(f.read(count * 10), dtype=np.dtype([("assocMsrCount", "<u4"), ("amlStnIndex", "<u4"), ("validity", "<u2")]))
        return raw


def read_seg(path):
    """returns (ISL, JSL, CML, net_ids): fixed columns 0 / 16 / 32 of the per-block tables (seg_file.cpp:305-392)"""
    lines = open(path).read().split("\n")
    i = 0
    while not lines[i].startswith("No. blocks produced"):
        i += 1
    B = int(lines[i][35:])
    i += 3
    counts = []
    for _ in range(B):
        t = lines[i].split()
        counts.append((int(t[1]), int(t[2]), int(t[3]), int(t[4])))  # net, junction, inner, msr
        i += 1
    ISL, JSL, CML = [], [], []
    for b in range(B):
        while not lines[i].startswith("Block %d" % (b + 1)):
            i += 1
        i += 9
        net, nj, ni, nm = counts[b]
        isl, jsl, cml = [], [], []
        c = 0
        while not lines[i].startswith("--------------------"):
            ln = lines[i]
            if c < ni:
                isl.append(int(ln[0:16]))
            if c < nj:
                jsl.append(int(ln[16:32]))
            if c < nm:
                cml.append(int(ln[32:48]))
            c += 1
            i += 1
        ISL.append(np.array(isl, dtype=np.uint32))
        JSL.append(np.array(jsl, dtype=np.uint32))
        CML.append(np.array(cml, dtype=np.uint32))
    return ISL, JSL, CML, np.array([c[0] for c in counts], dtype=np.uint32)
