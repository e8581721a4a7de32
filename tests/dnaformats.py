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


# ---- writers (tests only): used to build networks the synthetic generator cannot make ----
def _header(app=b"   PYTESTS"):
    return b"VERSION   " + b"       1.2" + b"CREATED ON" + b"2026-09-28" + b"CREATED BY" + app[:10].rjust(10)


def _meta(count):
    out = struct.pack("<Q", count) + b"\x00" + b"pytest".ljust(20, b"\x00") + b"7843".ljust(7, b"\x00")
    out += b"01.01.2020".ljust(12, b"\x00") * 2 + b"\x00\x00" + struct.pack("<Q", 0) + struct.pack("<Q", 0)
    return out


def write_bst(path, stations):
    with open(path, "wb") as f:
        f.write(_header() + _meta(len(stations)) + np.ascontiguousarray(stations).tobytes())


def write_bms(path, msrs):
    with open(path, "wb") as f:
        f.write(_header() + _meta(len(msrs)) + np.ascontiguousarray(msrs).tobytes())


def write_asl(path, counts):
    with open(path, "wb") as f:
        f.write(_header() + struct.pack("<Q", len(counts)))
        off = 0
        for c in counts:
            f.write(struct.pack("<IIH", int(c), off, 1))
            off += int(c)


def write_seg(path, ISL, JSL, CML, nets, bms):
    """same fixed-column layout as SegFile::WriteSegFile (seg_file.cpp:590-721)"""
    L = []
    rule80 = "-" * 80
    L += [rule80, "DYNADJUST SEGMENTATION OUTPUT FILE", "", "Version:".ljust(35) + "tests", "Build:".ljust(35) + "tests",
          "File created:".ljust(35) + "now", "File name:".ljust(35) + path, "", "Command line arguments: ".ljust(35) + "tests", "",
          "Stations file:".ljust(35) + "x.bst", "Measurements file:".ljust(35) + "x.bms", "",
          "Minimum inner stations".ljust(35) + "0", "Block size threshold".ljust(35) + "0", "Starting station(s)".ljust(35) + " ",
          rule80, "", "SEGMENTATION SUMMARY".ljust(35), "", "No. blocks produced".ljust(35) + str(len(ISL))]
    rule = "-" * 90
    L += [rule, "  Block".ljust(14) + "Network ID".ljust(14) + "Junction stns".ljust(16) + "Inner stns".ljust(16) + "Measurements".ljust(16) + "Total stns".ljust(16)]
    for b in range(len(ISL)):
        L.append("  " + str(b + 1).ljust(12) + str(int(nets[b])).ljust(14) + str(len(JSL[b])).ljust(16) + str(len(ISL[b])).ljust(16) +
                 str(len(CML[b])).ljust(16) + str(len(ISL[b]) + len(JSL[b])).ljust(16))
    L += [rule, "", "INDIVIDUAL BLOCK DATA", rule]
    br = "-" * 53
    for b in range(len(ISL)):
        L += ["", "Block %d" % (b + 1), br, "Junction stns:".ljust(16) + str(len(JSL[b])), "Inner stns:".ljust(16) + str(len(ISL[b])),
              "Measurements:".ljust(16) + str(len(CML[b])), "Total stns:".ljust(16) + str(len(ISL[b]) + len(JSL[b])), "",
              "Inner stns".ljust(16) + "Junction stns".ljust(16) + "Measurements".ljust(16) + "Type".ljust(5), br]
        rows = max(len(ISL[b]), len(JSL[b]), len(CML[b]), 1)
        for r in range(rows):
            s = (str(int(ISL[b][r])) if r < len(ISL[b]) else " ").ljust(16)
            s += (str(int(JSL[b][r])) if r < len(JSL[b]) else " ").ljust(16)
            if r < len(CML[b]):
                s += str(int(CML[b][r])).ljust(16) + bms["measType"][int(CML[b][r])].decode().ljust(5)
            L.append(s)
        L.append(br)
    L.append("")
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


def merge_networks(bases, out_base):
    """Concatenate independently generated networks into one multi-network project: station and
    measurement indices are offset, block lists are appended and every source keeps its own
    contiguous-network id (so its first/last/isolated block flags follow dnaadjust.cpp:10449-10474)."""
    bst_all, bms_all, asl_all, ISL, JSL, CML, nets = [], [], [], [], [], [], []
    s_off = m_off = 0
    for net_id, base in enumerate(bases):
        bst, bms, asl = read_bst(base + ".bst").copy(), read_bms(base + ".bms").copy(), read_asl(base + ".asl")
        bms["station1"] += s_off
        bms["station2"] += s_off
        i, j, c, _ = read_seg(base + ".seg")
        ISL += [x + s_off for x in i]
        JSL += [x + s_off for x in j]
        CML += [x + m_off for x in c]
        nets += [net_id] * len(i)
        bst_all.append(bst)
        bms_all.append(bms)
        asl_all += list(asl["assocMsrCount"])
        s_off += len(bst)
        m_off += len(bms)
    # concatenate through raw bytes: np.concatenate would repack the explicit-offset dtypes
    def cat(parts, dt):
        raw = np.concatenate([np.frombuffer(np.ascontiguousarray(p).tobytes(), dtype=np.uint8) for p in parts])
        return np.frombuffer(raw.tobytes(), dtype=dt)
    bst = cat(bst_all, STATION_DT)
    bms = cat(bms_all, MEASUREMENT_DT)
    write_bst(out_base + ".bst", bst)
    write_bms(out_base + ".bms", bms)
    write_asl(out_base + ".asl", asl_all)
    write_seg(out_base + ".seg", ISL, JSL, CML, nets, bms)


def read_mtx(path, count):
    """matrix_2d binary stream records of <net>-rva.mtx / -pam.mtx (include/math/dnamatrix_contiguous.cpp:39-91): per matrix
    6 x u32 (type: 1 = packed lower, rows, cols, mem_rows, mem_cols, pad), the doubles, 2 x u32 footer -> [(type, rows, cols, data)]"""
    import struct
    out = []
    with open(path, "rb") as f:
        for _ in range(count):
            mtype, rows, cols, mrows, mcols, pad = struct.unpack("<6I", f.read(24))
            assert (mrows, mcols, pad) == (rows, cols, 0)
            n = rows * (rows + 1) // 2 if mtype == 1 else rows * cols
            data = np.frombuffer(f.read(8 * n), dtype=np.float64)
            assert struct.unpack("<2I", f.read(8)) == (0, 0)
            out.append((mtype, rows, cols, data))
        assert f.read() == b""
    return out
