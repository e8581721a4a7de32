"""ctypes wrapper of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
MKL = "/opt/conda/lib/libmkl_rt.so"

f64p = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)


class OrcNetwork(C.Structure):
    _fields_ = [("n_stations", C.c_uint32), ("xyz0", f64p), ("constraints", C.c_char_p), ("n_baselines", C.c_uint32),
                ("stn1", u32p), ("stn2", u32p), ("obs", f64p), ("vcv6", f64p), ("n_blocks", C.c_uint32),
                ("isl_off", u32p), ("isl", u32p), ("jsl_off", u32p), ("jsl", u32p), ("cml_off", u32p), ("cml", u32p),
                ("net_id", u32p), ("n_clusters", C.c_uint32), ("cluster_off", u32p), ("cluster_vcv", f64p),
                ("n_tmsr", C.c_uint32), ("t_type", C.c_char_p), ("t_stn", u32p), ("t_value", f64p), ("t_var", f64p),
                ("t_ih", f64p), ("t_th", f64p), ("stn_llh", f64p), ("stn_geoid", f64p), ("stn_defl", f64p),
                ("n_dsets", C.c_uint32), ("dset_first", u32p), ("dset_size", u32p), ("dset_w", f64p), ("stn_type", C.POINTER(C.c_uint16))]


class OrcSettings(C.Structure):
    _fields_ = [("fixed_std_dev", C.c_double), ("free_std_dev", C.c_double), ("iteration_threshold", C.c_double),
                ("max_iterations", C.c_uint32), ("scale_normals_to_unity", C.c_int), ("threads", C.c_int)]


class OrcOscRecord(C.Structure):
    _fields_ = [("station", C.c_uint32), ("first_iteration", C.c_uint32), ("last_iteration", C.c_uint32), ("cycles", C.c_uint32),
                ("first_mag", C.c_double), ("last_mag", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("cz", C.c_double)]


class OrcStatistics(C.Structure):
    _fields_ = [("chi_squared", C.c_double), ("sigma_zero", C.c_double), ("global_pelzer", C.c_double),
                ("measurement_params", C.c_uint32), ("unknown_params", C.c_uint32), ("potential_outliers", C.c_uint32),
                ("dof", C.c_int)]


TERRESTRIAL_TYPES = b"ABCEHIJKLMPQRSVZ"
TMSR_FIELDS = ("measAdj", "measCorr", "measAdjPrec", "residualPrec", "NStat", "PelzerRel", "measPrec", "preAdjCorr")
MSR_FIELDS = ("measAdj", "measCorr", "measAdjPrec", "residualPrec", "NStat", "PelzerRel", "measPrec")

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        lib = C.CDLL(LIB)
        lib.orc_packed_index.restype = C.c_size_t
        lib.orc_packed_index.argtypes = [C.c_uint32] * 3
        lib.orc_set_lapack.argtypes = [C.c_char_p]
        lib.orc_lapack_name.restype = C.c_char_p
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_set_threads_local.argtypes = [C.c_int]
        lib.orc_adjust_forward_pass.argtypes = [C.c_void_p]
        lib.orc_adjust_reverse_pass.argtypes = [C.c_void_p]
        lib.orc_potrf_lower.argtypes = [C.c_uint32, f64p, C.c_uint32]
        lib.orc_potri_lower.argtypes = [C.c_uint32, f64p, C.c_uint32]
        lib.orc_cholesky_inverse_packed.argtypes = [f64p, C.c_uint32]
        lib.orc_inverse_normals_packed.argtypes = [f64p, C.c_uint32, C.c_int]
        lib.orc_cholesky_inverse_full.argtypes = [f64p, C.c_uint32, C.c_uint32]
        lib.orc_scale_symmetric_diagonal_packed.argtypes = [f64p, C.c_uint32, f64p]
        lib.orc_multiply_sym_packed.argtypes = [f64p, f64p, f64p, C.c_uint32]
        lib.orc_geo_to_cart.argtypes = [C.c_double] * 3 + [f64p] * 3
        lib.orc_weight_3x3.argtypes = [f64p, f64p]
        lib.orc_propagate_geo_cart.argtypes = [f64p, C.c_uint32, f64p, C.c_int]
        lib.orc_scale_gps_vcv.argtypes = [f64p, C.c_uint32, f64p, C.c_double, C.c_double, C.c_double, C.c_int]
        lib.orc_adjust_oscillation_history.restype = C.c_uint32
        lib.orc_adjust_oscillation_history.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        lib.orc_adjust_create.restype = C.c_void_p
        lib.orc_adjust_create.argtypes = [C.POINTER(OrcNetwork), C.POINTER(OrcSettings), C.c_int]
        lib.orc_adjust_destroy.argtypes = [C.c_void_p]
        lib.orc_adjust_prepare.argtypes = [C.c_void_p]
        lib.orc_adjust_run.argtypes = [C.c_void_p]
        lib.orc_adjust_run_block1.argtypes = [C.c_void_p]
        lib.orc_adjust_iteration.argtypes = [C.c_void_p]
        lib.orc_adjust_iterations.restype = C.c_uint32
        lib.orc_adjust_iterations.argtypes = [C.c_void_p]
        lib.orc_adjust_max_correction.restype = C.c_double
        lib.orc_adjust_max_correction.argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_adjust_block_unknowns.restype = C.c_uint32
        lib.orc_adjust_block_unknowns.argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_adjust_block_stations.restype = u32p
        lib.orc_adjust_block_stations.argtypes = [C.c_void_p, C.c_uint32, u32p]
        for nm in ("orc_adjust_block_estimates", "orc_adjust_block_variances", "orc_adjust_block_normals"):
            getattr(lib, nm).restype = f64p
            getattr(lib, nm).argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_adjust_block_b.restype = f64p
        lib.orc_adjust_block_b.argtypes = [C.c_void_p, C.c_uint32, u32p]
        lib.orc_adjust_weights.restype = f64p
        lib.orc_adjust_weights.argtypes = [C.c_void_p]
        lib.orc_adjust_error.restype = C.c_char_p
        lib.orc_adjust_error.argtypes = [C.c_void_p]
        lib.orc_adjust_solve_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), f64p]
        lib.orc_adjust_statistics.argtypes = [C.c_void_p, C.c_double, C.POINTER(OrcStatistics)]
        lib.orc_adjust_msr_field.restype = f64p
        lib.orc_adjust_msr_field.argtypes = [C.c_void_p, C.c_int]
        lib.orc_adjust_tmsr_field.restype = f64p
        lib.orc_adjust_tmsr_field.argtypes = [C.c_void_p, C.c_int]
        lib.orc_adjust_station_llh.restype = f64p
        lib.orc_adjust_station_llh.argtypes = [C.c_void_p]
        lib.orc_tmsr_evaluate.argtypes = [C.c_void_p, C.c_uint32, f64p, C.POINTER(C.c_double), f64p]
        lib.orc_adjust_block_prec_adj_msrs.restype = f64p
        lib.orc_adjust_block_prec_adj_msrs.argtypes = [C.c_void_p, C.c_uint32, u32p]
        _lib = lib
    return _lib


def scipy_openblas_path():
    """the OpenBLAS inside the scipy wheel (LP64, symbols prefixed scipy_): a second host LAPACK for the CPU baseline's probe"""
    import glob
    try:
        import scipy
    except ImportError:
        return None
    hits = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas-*.so"))
    return hits[0] if hits else None


def use_lapack(path):
    lib = load()
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    if path and os.path.exists(path):
        return lib.orc_set_lapack(path.encode()) == 0
    lib.orc_set_lapack(None)
    return False


def use_mkl(enable=True):
    """switch the oracle's dpotrf/dpotri to the MKL runtime (the LAPACK the reference links)"""
    lib = load()
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")   # see tests/conftest.py
    if enable and os.path.exists(MKL):
        return lib.orc_set_lapack(MKL.encode()) == 0
    lib.orc_set_lapack(None)
    return False


def _p(a, t):
    return a.ctypes.data_as(t)


def cholesky_inverse_packed(ap, n, scale=False):
    lib = load()
    ap = np.array(ap, dtype=np.float64, copy=True)
    info = lib.orc_inverse_normals_packed(_p(ap, f64p), n, int(scale))
    return ap, info


def multiply_sym_packed(ap, x, n):
    lib = load()
    ap = np.ascontiguousarray(ap, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(n)
    lib.orc_multiply_sym_packed(_p(ap, f64p), _p(x, f64p), _p(y, f64p), n)
    return y


def geo_to_cart(lat, lon, h):
    lib = load()
    x, y, z = C.c_double(), C.c_double(), C.c_double()
    lib.orc_geo_to_cart(lat, lon, h, C.byref(x), C.byref(y), C.byref(z))
    return x.value, y.value, z.value


def propagate_geo_cart(V, llh, geo_to_cart=True):
    """PropagateVariances_GeoCart_Cluster on a 3k x 3k matrix (numpy, any order); llh = k x 3 (lat, lon, h)"""
    lib = load()
    k = V.shape[0] // 3
    v = np.asfortranarray(V, dtype=np.float64).copy(order="F")
    p = np.ascontiguousarray(llh, dtype=np.float64)
    lib.orc_propagate_geo_cart(_p(v, f64p), k, _p(p, f64p), int(geo_to_cart))
    return v


def scale_gps_vcv(V, llh, pscale, lscale, hscale, v_is_geographic=False):
    """ScaleGPSVCV_Cluster"""
    lib = load()
    k = V.shape[0] // 3
    v = np.asfortranarray(V, dtype=np.float64).copy(order="F")
    p = np.ascontiguousarray(llh, dtype=np.float64)
    lib.orc_scale_gps_vcv(_p(v, f64p), k, _p(p, f64p), float(pscale), float(lscale), float(hscale), int(v_is_geographic))
    return v


def weight_3x3(v6):
    lib = load()
    v = np.ascontiguousarray(v6, dtype=np.float64)
    w = np.empty(6)
    rc = lib.orc_weight_3x3(_p(v, f64p), _p(w, f64p))
    return w, rc


class Network:
    """Arrays of a GNSS network read with tests/dnaformats.py, in the form the oracle takes."""

    def __init__(self, base, phased):
        from . import dnaformats as F
        bst = F.read_bst(base + ".bst")
        bms = F.read_bms(base + ".bms")
        self.n_stations = len(bst)
        self.xyz0 = np.empty(3 * self.n_stations)
        for s in range(self.n_stations):
            self.xyz0[3 * s:3 * s + 3] = geo_to_cart(float(bst["currentLatitude"][s]), float(bst["currentLongitude"][s]),
                                                     float(bst["currentHeight"][s]))
        self.constraints = b"".join(bytes(c[:3]).ljust(3, b"F") for c in bst["stationConst"])
        self.n_clusters = 0
        self.cluster_off = np.zeros(1, dtype=np.uint32)
        self.cluster_vcv = np.zeros(1)
        self._llh = np.stack([bst["currentLatitude"], bst["currentLongitude"], bst["currentHeight"]], axis=1).astype(np.float64)
        self._geoid = np.ascontiguousarray(bst["geoidSep"], dtype=np.float64)
        self._supplied_type = np.ascontiguousarray(bst["suppliedStationType"], dtype=np.uint16)
        self._vdef = np.ascontiguousarray(bst["verticalDef"], dtype=np.float64)
        self._mdef = np.ascontiguousarray(bst["meridianDef"], dtype=np.float64)
        unit = lambda a: np.where(np.asarray(a) < 1e-6, 1.0, a)
        partial = any(np.any(np.abs(unit(bms[k]) - 1.0) > 1e-5) for k in ("scale1", "scale2", "scale3"))
        terr = np.isin(bms["measType"], [bytes([c]) for c in TERRESTRIAL_TYPES + b"D"])
        self.n_tmsr = 0
        if np.all(bms["measType"] == b"G") and not partial and not np.any(terr):
            starts = np.nonzero((bms["measStart"] == 0) & (~bms["ignore"]))[0]
            self.bl_of_record = {int(m): i for i, m in enumerate(starts)}
            self.n_baselines = len(starts)
            self.stn1 = np.ascontiguousarray(bms["station1"][starts], dtype=np.uint32)
            self.stn2 = np.ascontiguousarray(bms["station2"][starts], dtype=np.uint32)
            self.obs = np.ascontiguousarray(np.stack([bms["term1"][starts], bms["term1"][starts + 1], bms["term1"][starts + 2]], axis=1)).ravel()
            vs = np.where(bms["scale4"][starts] < 1e-6, 1.0, bms["scale4"][starts])
            v6 = np.stack([bms["term2"][starts], bms["term2"][starts + 1], bms["term3"][starts + 1],
                           bms["term2"][starts + 2], bms["term3"][starts + 2], bms["term4"][starts + 2]], axis=1)
            scale = np.where(np.abs(vs - 1.0) > 1e-5, vs, 1.0)
            self.vcv6 = np.ascontiguousarray(v6 * scale[:, None]).ravel()
        else:
            self._parse_clusters(bms)
        if phased:
            ISL, JSL, CML, nets = F.read_seg(base + ".seg")
            self.n_blocks = len(ISL)
            self.isl_off, self.isl = self._csr(ISL)
            self.jsl_off, self.jsl = self._csr(JSL)
            # (a direction set is one measurement of k angles: k consecutive entries)
            cml_bl = [np.array([e for m in c for e in self._entries(int(m))], dtype=np.uint32) for c in CML]
            self.cml_off, self.cml = self._csr(cml_bl)
            self.net_id = np.ascontiguousarray(nets, dtype=np.uint32)
        else:
            self.n_blocks = 1
            z = np.zeros(2, dtype=np.uint32)
            self.isl_off = self.jsl_off = self.cml_off = z
            self.isl = self.jsl = self.cml = np.zeros(1, dtype=np.uint32)
            self.net_id = np.zeros(1, dtype=np.uint32)
            if self.n_tmsr:
                # record order, like BuildSimultaneousLists of the facade
                order = [e for r in sorted(self.bl_of_record) for e in self._entries(r)]
                self.cml = np.asarray(order, dtype=np.uint32)
                self.cml_off = np.asarray([0, len(order)], dtype=np.uint32)

    def _entries(self, record):
        e = self.bl_of_record[record]
        return list(range(e[0], e[0] + e[1])) if isinstance(e, tuple) else [e]

    def _parse_clusters(self, bms):
        """G / X / Y records -> vectors + clusters: the .bms layout and the scaling / frame rules of
        LoadVarianceScaling, LoadVarianceMatrix_G/_X/_Y (dnaadjust.cpp:4453, 4214, 4312, 4494) and, for point clusters
        given in latitude / longitude / height, UpdateDesignNormalMeasMatrices_Y (dnaadjust.cpp:6281-6318)"""
        stn1, stn2, obs, vcv, off = [], [], [], [], [0]
        self.bl_of_record = {}
        i, n = 0, len(bms)
        tiny = 1e-6                                   # min(PRECISION_1E5, fixed_std_dev)
        tm = {"type": [], "stn": [], "value": [], "var": [], "ih": [], "th": [], "record": []}
        dsets, dset_records = [], {}
        while i < n:
            t = bytes(bms["measType"][i])
            if t == b"D":
                # direction set (UpdateDesignNormalMeasMatrices_D dnaadjust.cpp:5082, LoadVarianceMatrix_D :4059): the record of the
                # reference direction + vectorCount1 - 1 direction records; angles between consecutive non-ignored directions,
                # variance matrix of the differences of independent directions, derived values stored with the later direction
                assert bms["measStart"][i] == 0
                total = int(bms["vectorCount1"][i])
                if bms["ignore"][i]:
                    i += total
                    continue
                recs = [i] + [j for j in range(i + 1, i + total) if not bms["ignore"][j]]
                k = len(recs) - 1
                assert k >= 1 and int(bms["vectorCount2"][i]) == k + 1
                first = len(tm["type"])
                V = np.zeros((k, k))
                for a in range(k):
                    ra, rb = recs[a], recs[a + 1]
                    ang = float(bms["term1"][rb]) - float(bms["term1"][ra])
                    if ang < 0:
                        ang += 2 * np.pi
                    if ang > 2 * np.pi:
                        ang -= 2 * np.pi
                    V[a, a] = float(bms["term2"][ra]) + float(bms["term2"][rb])
                    if a + 1 < k:
                        V[a, a + 1] = V[a + 1, a] = -float(bms["term2"][rb])
                    tm["type"].append(b"D")
                    tm["stn"].append([int(bms["station1"][i]), int(bms["station2"][ra]), int(bms["station2"][rb])])
                    tm["value"].append(ang)
                    tm["var"].append(V[a, a])
                    tm["ih"].append(float(bms["term3"][ra]))     # the angle's record is a copy of the earlier direction's
                    tm["th"].append(float(bms["term4"][ra]))
                    tm["record"].append(rb)
                dsets.append((first, k, np.linalg.inv(V)))
                dset_records[i] = (first, k)
                i += total
                continue
            if t[0] in TERRESTRIAL_TYPES:
                assert bms["measStart"][i] == 0
                if bms["ignore"][i]:                  # an ignored measurement is in no block's measurement list
                    i += 1
                    continue
                tm["type"].append(t)
                tm["stn"].append([int(bms["station1"][i]), int(bms["station2"][i]), int(bms["station3"][i])])
                tm["value"].append(float(bms["term1"][i]))
                tm["var"].append(float(bms["term2"][i]))
                tm["ih"].append(float(bms["term3"][i]))
                tm["th"].append(float(bms["term4"][i]))
                tm["record"].append(i)
                i += 1
                continue
            assert t in (b"G", b"X", b"Y") and bms["measStart"][i] == 0, t
            if bms["ignore"][i]:
                for _ in range(1 if t == b"G" else int(bms["vectorCount1"][i])):
                    i += 3 + (0 if t == b"G" else 3 * int(bms["vectorCount2"][i]))
                continue
            self.bl_of_record[i] = len(off) - 1      # cml entries become cluster indices
            k = 1 if t == b"G" else int(bms["vectorCount1"][i])
            unit = lambda v: 1.0 if v < tiny else float(v)
            vs, ps, ls, hs = (unit(float(bms[f][i])) for f in ("scale4", "scale1", "scale2", "scale3"))
            scale_matrix = abs(vs - 1.0) > 1e-5
            scale_partial = abs(ps - 1.0) > 1e-5 or abs(ls - 1.0) > 1e-5 or abs(hs - 1.0) > 1e-5
            if scale_partial and scale_matrix:
                ps, ls, hs = ps * vs, ls * vs, hs * vs
            ctype = bytes(bms["coordType"][i]).rstrip(b"\x00").strip()
            geographic = t == b"Y" and ctype in (b"LLH", b"LLh")
            V = np.zeros((3 * k, 3 * k))
            pos = np.zeros((k, 3))
            for j in range(k):
                r0 = 3 * j
                o = [float(bms["term1"][i]), float(bms["term1"][i + 1]), float(bms["term1"][i + 2])]
                s1 = int(bms["station1"][i])
                pos[j] = self._llh[s1]
                if geographic:
                    hgt = o[2]
                    if ctype == b"LLH" and abs(self._geoid[s1]) > 1e-4:
                        hgt += float(self._geoid[s1])
                    o = list(geo_to_cart(o[0], o[1], hgt))
                obs += o
                if t == b"Y":
                    stn1.append(0xffffffff)
                    stn2.append(s1)
                else:
                    stn1.append(s1)
                    stn2.append(int(bms["station2"][i]))
                V[r0, r0] = bms["term2"][i]
                V[r0, r0 + 1] = bms["term2"][i + 1]
                V[r0 + 1, r0 + 1] = bms["term3"][i + 1]
                V[r0, r0 + 2] = bms["term2"][i + 2]
                V[r0 + 1, r0 + 2] = bms["term3"][i + 2]
                V[r0 + 2, r0 + 2] = bms["term4"][i + 2]
                ncov = 0 if t == b"G" else int(bms["vectorCount2"][i])
                i += 3
                for c in range(ncov):
                    c0 = 3 * (j + 1 + c)
                    for r in range(3):
                        V[r0 + r, c0:c0 + 3] = [bms["term1"][i + r], bms["term2"][i + r], bms["term3"][i + r]]
                    i += 3
            V = np.triu(V) + np.triu(V, 1).T
            if t != b"Y":
                if scale_matrix:
                    V = V * vs                        # "on the fly" (dnaadjust.cpp:4236, 4360)
                if scale_partial:
                    V = scale_gps_vcv(V, pos, ps, ls, hs, False)
            else:
                if scale_partial:
                    V = scale_gps_vcv(V, pos, ps, ls, hs, geographic)
                elif geographic:
                    V = propagate_geo_cart(V, pos, True)
                if scale_matrix and not scale_partial:
                    V = V * vs                        # dnaadjust.cpp:4650
            vcv.append(np.asarray(V).ravel(order="F"))
            off.append(len(stn1))
        self.n_baselines = len(stn1)
        self.stn1 = np.asarray(stn1, dtype=np.uint32)
        self.stn2 = np.asarray(stn2, dtype=np.uint32)
        self.obs = np.asarray(obs, dtype=np.float64)
        self.vcv6 = np.zeros(6 * self.n_baselines)
        self.n_clusters = len(off) - 1
        self.cluster_off = np.asarray(off, dtype=np.uint32)
        self.cluster_vcv = np.ascontiguousarray(np.concatenate(vcv)) if vcv else np.zeros(1)
        # terrestrial measurements: cml entry = n_clusters + index
        self.n_tmsr = len(tm["type"])
        if self.n_tmsr:
            assert self.n_clusters > 0 or self.n_baselines == 0
            self.t_type = b"".join(tm["type"])
            self.t_stn = np.asarray(tm["stn"], dtype=np.uint32).ravel()
            self.t_value = np.asarray(tm["value"])
            self.t_var = np.asarray(tm["var"])
            self.t_ih = np.asarray(tm["ih"])
            self.t_th = np.asarray(tm["th"])
            self.t_record = np.asarray(tm["record"])
            for q, r in enumerate(tm["record"]):
                if tm["type"][q] != b"D":
                    self.bl_of_record[r] = self.n_clusters + q
            for r, (first, k) in dset_records.items():
                self.bl_of_record[r] = (self.n_clusters + first, k)
            self.dset_first = np.asarray([d[0] for d in dsets] if dsets else [0], dtype=np.uint32)
            self.dset_size = np.asarray([d[1] for d in dsets] if dsets else [0], dtype=np.uint32)
            self.dset_w = np.ascontiguousarray(np.concatenate([d[2].ravel(order="F") for d in dsets])) if dsets else np.zeros(1)
            self.n_dsets = len(dsets)
            if self.n_clusters == 0:
                # the oracle tells clusters from terrestrial entries by n_clusters (or n_baselines)
                self.cluster_off = np.zeros(1, dtype=np.uint32)

    @staticmethod
    def _csr(lists):
        off = np.zeros(len(lists) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(l) for l in lists])
        flat = np.concatenate([np.asarray(l, dtype=np.uint32) for l in lists] + [np.zeros(0, dtype=np.uint32)]) if lists else np.zeros(0, np.uint32)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.uint32)
        return off, np.ascontiguousarray(flat, dtype=np.uint32)

    def c_struct(self):
        n = OrcNetwork()
        n.n_stations = self.n_stations
        n.xyz0 = _p(self.xyz0, f64p)
        n.constraints = self.constraints
        n.n_baselines = self.n_baselines
        n.stn1 = _p(self.stn1, u32p)
        n.stn2 = _p(self.stn2, u32p)
        n.obs = _p(self.obs, f64p)
        n.vcv6 = _p(self.vcv6, f64p)
        n.n_blocks = self.n_blocks
        n.isl_off, n.isl = _p(self.isl_off, u32p), _p(self.isl, u32p)
        n.jsl_off, n.jsl = _p(self.jsl_off, u32p), _p(self.jsl, u32p)
        n.cml_off, n.cml = _p(self.cml_off, u32p), _p(self.cml, u32p)
        n.net_id = _p(self.net_id, u32p)
        n.n_clusters = self.n_clusters
        n.cluster_off = _p(self.cluster_off, u32p)
        n.cluster_vcv = _p(self.cluster_vcv, f64p)
        n.n_tmsr = self.n_tmsr
        self._llh_flat = np.ascontiguousarray(self._llh).ravel()
        self._stn_type = np.ascontiguousarray(self._supplied_type, dtype=np.uint16)
        n.stn_llh = _p(self._llh_flat, f64p)
        n.stn_type = _p(self._stn_type, C.POINTER(C.c_uint16))
        if self.n_tmsr:
            n.n_dsets = getattr(self, "n_dsets", 0)
            if n.n_dsets:
                n.dset_first, n.dset_size, n.dset_w = _p(self.dset_first, u32p), _p(self.dset_size, u32p), _p(self.dset_w, f64p)
            self._defl = np.ascontiguousarray(np.stack([self._vdef, self._mdef], axis=1)).ravel()
            n.t_type = self.t_type
            n.t_stn, n.t_value, n.t_var = _p(self.t_stn, u32p), _p(self.t_value, f64p), _p(self.t_var, f64p)
            n.t_ih, n.t_th = _p(self.t_ih, f64p), _p(self.t_th, f64p)
            n.stn_llh, n.stn_geoid, n.stn_defl = _p(self._llh_flat, f64p), _p(self._geoid, f64p), _p(self._defl, f64p)
        return n


class Adjustment:
    """orc_adjustment: prepare() + run(), then per-block rigorous estimates / variances."""

    def __init__(self, net, phased, fixed_std_dev=1e-6, free_std_dev=10.0, iteration_threshold=float(np.float32(0.0005)),
                 max_iterations=10, scale_normals_to_unity=False, threads=0):
        self.lib = load()
        self.net = net
        self.cnet = net.c_struct()
        self.set = OrcSettings(fixed_std_dev, free_std_dev, iteration_threshold, max_iterations, int(scale_normals_to_unity), threads)
        self.h = self.lib.orc_adjust_create(C.byref(self.cnet), C.byref(self.set), int(phased))
        self.n_blocks = net.n_blocks if phased else 1

    def close(self):
        if self.h:
            self.lib.orc_adjust_destroy(self.h)
            self.h = None

    def prepare(self):
        rc = self.lib.orc_adjust_prepare(self.h)
        if rc:
            raise RuntimeError(self.lib.orc_adjust_error(self.h).decode())

    def run(self):
        st = self.lib.orc_adjust_run(self.h)
        if st == 5:
            raise RuntimeError(self.lib.orc_adjust_error(self.h).decode())
        return st

    def run_block1(self):
        """Phased_Block_1Mode (AdjustPhasedBlock1)"""
        st = self.lib.orc_adjust_run_block1(self.h)
        if st == 5:
            raise RuntimeError(self.lib.orc_adjust_error(self.h).decode())
        return st

    def statistics(self, confidence_interval=95.0):
        """GenerateStatistics: returns (OrcStatistics, {field: array of 3 per vector, network vector order})"""
        from scipy.stats import norm
        conf = confidence_interval * 0.01
        crit = float(norm.ppf(conf + (1.0 - conf) / 2.0))        # dnaadjust.cpp:203-206
        st = OrcStatistics()
        rc = self.lib.orc_adjust_statistics(self.h, crit, C.byref(st))
        assert rc == 0
        n = 3 * self.net.n_baselines
        fields = {nm: np.ctypeslib.as_array(self.lib.orc_adjust_msr_field(self.h, f), shape=(n,)).copy() for f, nm in enumerate(MSR_FIELDS)}
        return st, fields

    def tmsr_fields(self):
        n = self.net.n_tmsr
        return {nm: np.ctypeslib.as_array(self.lib.orc_adjust_tmsr_field(self.h, f), shape=(max(n, 1),))[:n].copy()
                for f, nm in enumerate(TMSR_FIELDS)}

    def station_llh(self):
        return np.ctypeslib.as_array(self.lib.orc_adjust_station_llh(self.h), shape=(self.net.n_stations, 3)).copy()

    def tmsr_evaluate(self, t, xyz9):
        x = np.ascontiguousarray(xyz9, dtype=np.float64)
        comp = C.c_double()
        row = np.zeros(9)
        self.lib.orc_tmsr_evaluate(self.h, t, _p(x, f64p), C.byref(comp), _p(row, f64p))
        return comp.value, row

    def block_prec_adj_msrs(self, b):
        rows = C.c_uint32()
        p = self.lib.orc_adjust_block_prec_adj_msrs(self.h, b, C.byref(rows))
        return np.ctypeslib.as_array(p, shape=(rows.value,)).copy()

    def oscillation_history(self):
        """UpdateIterationDiagnostics' records (ADJ:7450-7554), by station: dicts with the last correction in cartesian components"""
        n = self.lib.orc_adjust_oscillation_history(self.h, None, 0)
        recs = (OrcOscRecord * max(1, n))()
        self.lib.orc_adjust_oscillation_history(self.h, recs, n)
        out = [{"station": r.station, "first_iteration": r.first_iteration, "last_iteration": r.last_iteration, "cycles": r.cycles,
                "first_mag": r.first_mag, "last_mag": r.last_mag, "last_xyz": (r.cx, r.cy, r.cz)} for r in recs[:n]]
        return sorted(out, key=lambda d: d["station"])

    def iteration(self):
        if self.lib.orc_adjust_iteration(self.h):
            raise RuntimeError(self.lib.orc_adjust_error(self.h).decode())

    def iterations(self):
        return self.lib.orc_adjust_iterations(self.h)

    def max_correction(self, it):
        return self.lib.orc_adjust_max_correction(self.h, it)

    def block_stations(self, b):
        cnt = C.c_uint32()
        p = self.lib.orc_adjust_block_stations(self.h, b, C.byref(cnt))
        return np.ctypeslib.as_array(p, shape=(cnt.value,)).copy()

    def _vec(self, fn, b, count):
        p = fn(self.h, b)
        return np.ctypeslib.as_array(p, shape=(count,)).copy()

    def block_estimates(self, b):
        n = self.lib.orc_adjust_block_unknowns(self.h, b)
        return self._vec(self.lib.orc_adjust_block_estimates, b, n)

    def block_variances(self, b):
        n = self.lib.orc_adjust_block_unknowns(self.h, b)
        return self._vec(self.lib.orc_adjust_block_variances, b, n * (n + 1) // 2)

    def block_normals(self, b):
        n = self.lib.orc_adjust_block_unknowns(self.h, b)
        return self._vec(self.lib.orc_adjust_block_normals, b, n * (n + 1) // 2)

    def block_b(self, b):
        rows = C.c_uint32()
        p = self.lib.orc_adjust_block_b(self.h, b, C.byref(rows))
        return np.ctypeslib.as_array(p, shape=(rows.value,)).copy()

    def weights(self):
        p = self.lib.orc_adjust_weights(self.h)
        return np.ctypeslib.as_array(p, shape=(6 * self.net.n_baselines,)).copy()

    def solve_stats(self):
        s = C.c_uint64()
        f = C.c_double()
        self.lib.orc_adjust_solve_stats(self.h, C.byref(s), C.byref(f))
        return s.value, f.value
