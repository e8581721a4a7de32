"""The on-disk boundary: the product's C++ writers/readers (.bst/.bms/.asl/.seg) against an
independent numpy reader with the documented byte offsets, plus the committed tiny network."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import dnaformats as F


def test_record_sizes(built):
    assert built.dnaio_sizeof_station() == 352 == F.STATION_DT.itemsize      # dnatypes-structs.hpp:270-323
    assert built.dnaio_sizeof_measurement() == 208 == F.MEASUREMENT_DT.itemsize   # dnameasurement.hpp:133-194


def _summary(lib, base):
    ns, nm, na = C.c_uint64(), C.c_uint64(), C.c_uint64()
    err = C.create_string_buffer(512)
    rc = lib.dnaio_file_summary((base + ".bst").encode(), (base + ".bms").encode(), (base + ".asl").encode(),
                                C.byref(ns), C.byref(nm), C.byref(na), err, 512)
    assert rc == 0, err.value
    return ns.value, nm.value, na.value


def test_golden_network_files(built, golden_dir):
    base = os.path.join(golden_dir, "tiny_net")
    ns, nm, na = _summary(built, base)
    bst, bms, asl = F.read_bst(base + ".bst"), F.read_bms(base + ".bms"), F.read_asl(base + ".asl")
    assert (ns, nm, na) == (len(bst), len(bms), len(asl)) == (12, len(bms), 12)
    assert nm % 3 == 0
    assert set(bytes(c[:3]) for c in bst["stationConst"]) == {b"CCC", b"FFF"}
    assert np.all(bms["measType"] == b"G")
    assert list(bms["measStart"][:6]) == [0, 1, 2, 0, 1, 2]
    # a G baseline: X row holds sXX, Y row sXY,sYY, Z row sXZ,sYZ,sZZ (dnaadjust.cpp:4236-4249): PD matrix
    v = np.array([[bms["term2"][0], bms["term2"][1], bms["term2"][2]],
                  [bms["term2"][1], bms["term3"][1], bms["term3"][2]],
                  [bms["term2"][2], bms["term3"][2], bms["term4"][2]]])
    assert np.all(np.linalg.eigvalsh(v) > 0)


def test_seg_reader_matches_numpy_reader(built, golden_dir):
    base = os.path.join(golden_dir, "tiny_net")
    ISL, JSL, CML, nets = F.read_seg(base + ".seg")
    nb = C.c_uint32()
    out = (C.c_uint32 * (8 * 16))()
    err = C.create_string_buffer(512)
    rc = built.dnaio_seg_summary((base + ".seg").encode(), (base + ".bms").encode(), C.byref(nb), out, 16, err, 512)
    assert rc == 0, err.value
    assert nb.value == len(ISL) == 3
    for b in range(nb.value):
        o = out[8 * b:8 * b + 8]
        assert o[0] == nets[b] and o[1] == len(JSL[b]) and o[2] == len(ISL[b]) and o[3] == len(CML[b])
        assert o[4] == 3 * len(CML[b])            # design rows: 3 per G baseline (seg_file.cpp:355-359)
        assert o[5] == ISL[b][0]
        assert o[6] == (JSL[b][0] if len(JSL[b]) else 0xffffffff)
        assert o[7] == CML[b][0]
    # segmentation invariants the adjustment relies on (dnaadjust.cpp:1072, 7803)
    for b in range(len(ISL) - 1):
        nxt = set(ISL[b + 1]) | set(JSL[b + 1])
        assert set(JSL[b]) <= nxt
    allm = np.concatenate(CML)
    assert len(set(allm)) == len(allm)


def test_generator_round_trip(built, tmp_path):
    from dynadjust_amd import adjust
    info = adjust.write_synthetic_network(str(tmp_path), "g", 9, 7, 120, 4, seed=7)
    base = str(tmp_path / "g")
    assert _summary(built, base) == (63, 360, 63)
    assert info["baselines"] == 120 and info["blocks"] == 4
    bst, bms = F.read_bst(base + ".bst"), F.read_bms(base + ".bms")
    truth = np.fromfile(base + ".truth").reshape(-1, 3)
    # initial coordinates are the truth perturbed by ~5 cm per axis
    from tests import oracle
    xyz0 = np.array([oracle.geo_to_cart(float(s["currentLatitude"]), float(s["currentLongitude"]), float(s["currentHeight"])) for s in bst])
    d = np.abs(xyz0 - truth)
    assert 0.001 < d.max() < 0.5
    # observations are truth baselines + millimetre noise
    s1, s2 = bms["station1"][0::3], bms["station2"][0::3]
    obs = np.stack([bms["term1"][0::3], bms["term1"][1::3], bms["term1"][2::3]], axis=1)
    assert np.abs(obs - (truth[s2] - truth[s1])).max() < 0.05
    ISL, JSL, CML, nets = F.read_seg(base + ".seg")
    assert sorted(np.concatenate(ISL)) == list(range(63))
    assert sum(len(c) for c in CML) == 120
    # same seed -> same files (deterministic generator)
    adjust.write_synthetic_network(str(tmp_path), "g2", 9, 7, 120, 4, seed=7)
    assert open(base + ".bms", "rb").read()[60:] == open(str(tmp_path / "g2.bms"), "rb").read()[60:]


def test_missing_and_corrupt_files(built, tmp_path):
    err = C.create_string_buffer(512)
    rc = built.dnaio_file_summary(b"/nonexistent/x.bst", None, None, None, None, None, err, 512)
    assert rc != 0 and b"error was encountered when opening" in err.value
    p = tmp_path / "bad.seg"
    p.write_text("not a seg file\n")
    nb = C.c_uint32()
    rc = built.dnaio_seg_summary(str(p).encode(), None, C.byref(nb), None, 0, err, 512)
    assert rc != 0
    # a version 1.1 measurement file must be refused like the reference does (bms_file.cpp:150-156)
    old = tmp_path / "old.bms"
    old.write_bytes(b"VERSION          1.1CREATED ON2020-01-01CREATED BY    IMPORT" + b"\x00" * 200)
    rc = built.dnaio_file_summary(None, str(old).encode(), None, None, None, None, err, 512)
    assert rc != 0 and b"predates observation_epoch" in err.value


def test_utm_conversion_matches_the_reference_report(golden_dir):
    """tests/urban_net.py's MGA -> geographic conversion (Krueger series) against the reference's own (Redfearn, dnaimport): the
    horizontally constrained stations of the urban sample keep their supplied position, which the report prints as latitude /
    longitude to 1e-9 degrees"""
    import math
    from tests import urban_net as U, dnatext as T
    st = {s["name"]: s for s in U.read_stations(os.path.join(golden_dir, "urban-network.stn"))}
    txt = open(os.path.join(golden_dir, "urban.phased.adj.expected")).read().split("\n")
    i = next(n for n, l in enumerate(txt) if l.startswith("Adjusted Coordinates")) + 5
    seen = 0
    while txt[i].strip():
        f = txt[i][20:].split()
        if f[0] in ("CCC", "CCF"):
            s = st[txt[i][:20].strip()]
            lat, lon = math.radians(T.dms_to_deg(float(f[1]))), math.radians(T.dms_to_deg(float(f[2])))
            assert abs(s["lat"] - lat) * 6.4e6 < 2e-4 and abs(s["lon"] - lon) * 6.4e6 * math.cos(lat) < 2e-4
            seen += 1
        i += 1
    assert seen == 2


def test_unique_id_hand_off_over_tcp(built):
    """what dna_adjust::PrepareAdjustment does between processes before ncclCommInitRank (dist_comm.cpp tcp_share_unique_id): rank 0
    serves ranks 1 .. world - 1 once each on the rendezvous address; a connection that does not introduce itself as one of them (a
    port probe), a rank that asks twice and a client that arrives before the server do not disturb it"""
    import ctypes as C
    import socket
    import threading
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 4
    secret = bytes(range(128))
    got, errs = {}, []

    def rank(r, delay=0.0):
        time.sleep(delay)
        buf = C.create_string_buffer(secret if r == 0 else b"\0" * 128, 128)
        err = C.create_string_buffer(256)
        rc = built.dnaadj_debug_tcp_share_unique_id(r, world, buf, b"127.0.0.1", port, 30.0, err, 256)
        if rc != 0:
            errs.append((r, err.value.decode()))
        got[r] = buf.raw

    threads = [threading.Thread(target=rank, args=(1,))]            # rank 1 is early: it retries until rank 0 listens
    threads[0].start()
    time.sleep(0.3)
    threads += [threading.Thread(target=rank, args=(0,)), threading.Thread(target=rank, args=(2, 0.4)), threading.Thread(target=rank, args=(3, 0.8))]
    for t in threads[1:]:
        t.start()
    time.sleep(0.2)
    # a stray connection that says nothing useful, and one that poses as rank 2 a second time: neither uses up rank 3's place
    for payload in (b"GET / HTTP/1.0\r\n\r\n", None):
        for _ in range(20):
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=2)
                break
            except OSError:
                time.sleep(0.05)
        else:
            continue
        with s:
            if payload is None:
                import struct
                s.sendall(struct.pack("<ii", 0x444e4131, 2))
                assert len(s.recv(128, socket.MSG_WAITALL)) == 128
            else:
                s.sendall(payload)
                s.settimeout(1.0)
                try:
                    assert s.recv(16) == b""
                except (socket.timeout, ConnectionError):
                    pass
    for t in threads:
        t.join(timeout=60)
    assert not errs, errs
    assert all(got[r] == secret for r in range(world))
