"""Thin numpy-facing wrapper over the dnagpu C-ABI (one context per GPU)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DnaGpuError, c_f64p, c_u32p


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_f64p)


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(c_u32p)


def packed_index(n, i, j):
    """matrix_2d::packed_index (dnamatrix_contiguous.hpp:363), i >= j."""
    return j * n - j * (j - 1) // 2 + (i - j)


def pack_lower(full):
    """column-major packed lower triangle of a square numpy matrix."""
    n = full.shape[0]
    out = np.empty(n * (n + 1) // 2, dtype=np.float64)
    k = 0
    for j in range(n):
        out[k:k + n - j] = full[j:, j]
        k += n - j
    return out


def unpack_lower(ap, n, symmetric=True):
    full = np.zeros((n, n), dtype=np.float64)
    k = 0
    for j in range(n):
        full[j:, j] = ap[k:k + n - j]
        k += n - j
    if symmetric:
        full = full + np.tril(full, -1).T
    return full


class ChainSource(C.Structure):
    _fields_ = [("m", C.c_void_p), ("junction", C.c_int), ("pos", C.POINTER(C.c_uint32)), ("k", C.c_size_t)]


class ChainStep(C.Structure):
    """dnagpu_chain_step (include/dnagpu.h)"""
    _fields_ = [("n_stn", C.c_uint32), ("est_blk", C.POINTER(C.c_uint32)), ("est_idx", C.POINTER(C.c_uint32)), ("n_src", C.c_int),
                ("src", ChainSource * 3), ("con_stn", C.POINTER(C.c_uint32)), ("con_w9", C.POINTER(C.c_double)), ("n_con", C.c_size_t),
                ("keep", C.POINTER(C.c_uint32)), ("n_keep", C.c_size_t), ("out", C.c_void_p), ("out_junction", C.c_int), ("matrix_only", C.c_int)]


class Matrix:
    def __init__(self, ctx, n_max):
        self.ctx = ctx
        self.n_max = int(n_max)
        self.n = 0
        h = C.c_void_p()
        ctx._chk(ctx.lib.dnagpu_matrix_create(ctx.h, self.n_max, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.dnagpu_matrix_destroy(self.ctx.h, self.h)
            self.h = None

    def reset(self, n, chain=0):
        self.ctx._chk(self.ctx.lib.dnagpu_matrix_reset(self.ctx.h, chain, self.h, int(n)))
        self.n = int(n)

    def upload_packed(self, ap, n, chain=0):
        ap, p = _f64(ap)
        assert ap.size == n * (n + 1) // 2
        self.ctx._chk(self.ctx.lib.dnagpu_matrix_upload_packed(self.ctx.h, chain, self.h, p, int(n)))
        self.n = int(n)

    def download_packed(self, chain=0):
        out = np.empty(self.n * (self.n + 1) // 2, dtype=np.float64)
        self.ctx._chk(self.ctx.lib.dnagpu_matrix_download_packed(self.ctx.h, chain, self.h, out.ctypes.data_as(c_f64p)))
        return out

    def invert(self, scale_to_unity=False, chain=0):
        self.ctx._chk(self.ctx.lib.dnagpu_invert(self.ctx.h, chain, self.h, int(bool(scale_to_unity))))


class DeviceContext:
    """One dnagpu_ctx.  Raises DnaGpuError on every non-zero return code."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.dnagpu_create(int(device), C.byref(h))
        if rc != 0:
            raise DnaGpuError(rc, "dnagpu_create failed (no MI355X visible?)")
        self.h = h

    def close(self):
        if self.h:
            self.lib.dnagpu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise DnaGpuError(rc, self.lib.dnagpu_last_error(self.h).decode())

    def last_info(self):
        return self.lib.dnagpu_last_info(self.h)

    def sync(self):
        self._chk(self.lib.dnagpu_sync(self.h))

    # ---- L2 seam ---------------------------------------------------------
    def cholesky_inverse_packed(self, ap, n, scale_to_unity=False):
        ap = np.array(ap, dtype=np.float64, copy=True)
        self._chk(self.lib.dnagpu_cholesky_inverse_packed(self.h, ap.ctypes.data_as(c_f64p), int(n), int(bool(scale_to_unity))))
        return ap

    def multiply_sym_packed(self, ap, x, n):
        ap, pa = _f64(ap)
        x, px = _f64(x)
        y = np.empty(n, dtype=np.float64)
        self._chk(self.lib.dnagpu_multiply_sym_packed(self.h, pa, px, y.ctypes.data_as(c_f64p), int(n)))
        return y

    # ---- profiling ---------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.lib.dnagpu_profile_enable(self.h, int(bool(on))))

    def profile_reset(self):
        self._chk(self.lib.dnagpu_profile_reset(self.h))

    def profile_get(self):
        f = C.c_double()
        ms = C.c_double()
        n = C.c_uint64()
        self._chk(self.lib.dnagpu_profile_get(self.h, C.byref(f), C.byref(ms), C.byref(n)))
        return {"gemm_flops": f.value, "gemm_ms": ms.value, "launches": n.value}

    def matrix(self, n_max):
        return Matrix(self, n_max)

    # ---- blocks --------------------------------------------------------------
    def block_create(self, blk, n_stations, n_baselines):
        self._chk(self.lib.dnagpu_block_create(self.h, blk, int(n_stations), int(n_baselines)))

    def block_destroy(self, blk):
        self._chk(self.lib.dnagpu_block_destroy(self.h, blk))

    def block_set_stations(self, blk, xyz):
        xyz, p = _f64(xyz)
        self._chk(self.lib.dnagpu_block_set_stations(self.h, blk, p))

    def block_set_baselines(self, blk, stn1, stn2, obs, vcv6):
        s1, p1 = _u32(stn1)
        s2, p2 = _u32(stn2)
        o, po = _f64(obs)
        v, pv = _f64(vcv6)
        self._chk(self.lib.dnagpu_block_set_baselines(self.h, blk, p1, p2, po, pv))

    def block_get_stations(self, blk, which, n_stations, chain=0):
        out = np.empty(3 * n_stations, dtype=np.float64)
        self._chk(self.lib.dnagpu_block_get_stations(self.h, chain, blk, which, out.ctypes.data_as(c_f64p)))
        return out

    def block_put_stations(self, blk, which, xyz, chain=0):
        xyz, p = _f64(xyz)
        self._chk(self.lib.dnagpu_block_put_stations(self.h, chain, blk, which, p))

    def block_copy_stations(self, blk, dst, src, chain=0):
        self._chk(self.lib.dnagpu_block_copy_stations(self.h, chain, blk, dst, src))

    def block_compute_b(self, blk, chain=0):
        self._chk(self.lib.dnagpu_block_compute_b(self.h, chain, blk))

    def block_get_b(self, blk, n_baselines, chain=0):
        out = np.empty(3 * n_baselines, dtype=np.float64)
        self._chk(self.lib.dnagpu_block_get_b(self.h, chain, blk, out.ctypes.data_as(c_f64p)))
        return out

    def block_get_weights(self, blk, n_baselines, chain=0):
        out = np.empty(6 * n_baselines, dtype=np.float64)
        self._chk(self.lib.dnagpu_block_get_weights(self.h, chain, blk, out.ctypes.data_as(c_f64p)))
        return out

    def form_normals(self, blk, m, n_stations, chain=0):
        self._chk(self.lib.dnagpu_form_normals(self.h, chain, blk, m.h))
        m.n = 3 * int(n_stations)

    def add_diag3x3(self, m, stn, w9, sign=1, chain=0):
        s, ps = _u32(stn)
        w, pw = _f64(w9)
        self._chk(self.lib.dnagpu_add_diag3x3(self.h, chain, m.h, ps, pw, s.size, sign))

    def form_rhs(self, blk, chain=0):
        self._chk(self.lib.dnagpu_form_rhs(self.h, chain, blk))

    def solve_corrections(self, blk, m, chain=0):
        self._chk(self.lib.dnagpu_solve_corrections(self.h, chain, blk, m.h))

    def update_estimates(self, blk, chain=0):
        v = C.c_double()
        r = C.c_uint32()
        self._chk(self.lib.dnagpu_update_estimates(self.h, chain, blk, C.byref(v), C.byref(r)))
        return v.value, r.value

    def block_get_corrections(self, blk, n_stations, chain=0):
        out = np.empty(3 * n_stations, dtype=np.float64)
        self._chk(self.lib.dnagpu_block_get_corrections(self.h, chain, blk, out.ctypes.data_as(c_f64p)))
        return out

    def block_get_rhs(self, blk, n_stations, chain=0):
        out = np.empty(3 * n_stations, dtype=np.float64)
        self._chk(self.lib.dnagpu_block_get_rhs(self.h, chain, blk, out.ctypes.data_as(c_f64p)))
        return out

    # ---- chain plans (dnagpu_chain_plan_*): steps = dicts with n_stn, est (block, idx) or None, sources [(Matrix, junction, pos)],
    #      con (stn, w9) or None, keep, out (Matrix), out_junction ----
    def chain_plan_create(self, steps, batch_first, max_bytes=1e12):
        arr = (ChainStep * len(steps))()
        self._plan_keep = keep_alive = []
        for q, st in enumerate(steps):
            d = arr[q]
            d.n_stn = int(st["n_stn"])
            if st.get("est") is not None:
                eb, pb = _u32(st["est"][0])
                ei, pi = _u32(st["est"][1])
                keep_alive += [eb, ei]
                d.est_blk, d.est_idx = pb, pi
            d.n_src = len(st["sources"])
            for r, (m, junction, pos) in enumerate(st["sources"]):
                ps, pp = _u32(pos)
                keep_alive.append(ps)
                d.src[r].m, d.src[r].junction, d.src[r].pos, d.src[r].k = m.h, int(junction), pp, ps.size
            if st.get("con") is not None:
                cs, pcs = _u32(st["con"][0])
                cw, pcw = _f64(st["con"][1])
                keep_alive += [cs, cw]
                d.con_stn, d.con_w9, d.n_con = pcs, pcw, cs.size
            kp, pk = _u32(st["keep"])
            keep_alive.append(kp)
            d.keep, d.n_keep = pk, kp.size
            d.out, d.out_junction = st["out"].h, int(st["out_junction"])
        bf, pbf = _u32(batch_first)
        h = C.c_void_p()
        self._chk(self.lib.dnagpu_chain_plan_create(self.h, len(steps), C.cast(arr, C.c_void_p), bf.size - 1, pbf, float(max_bytes), C.byref(h)))
        return h

    def chain_plan_run(self, plan, batch, chain=0):
        self._chk(self.lib.dnagpu_chain_plan_run(self.h, chain, plan, int(batch)))

    def chain_plan_run_rhs(self, plan, lo, hi, chain=0):
        self._chk(self.lib.dnagpu_chain_plan_run_rhs(self.h, chain, plan, int(lo), int(hi)))

    def chain_plan_destroy(self, plan):
        self.lib.dnagpu_chain_plan_destroy(self.h, plan)

    def junction_payload_put(self, m, F, est, rhs=None, chain=0):
        """m <- full matrix F (both triangles), attached vector est, information form when rhs is given (dnagpu_junction_import)"""
        n = F.shape[0]
        npad = ((n + 127) // 128) * 128 if n else 128
        buf = np.zeros(npad * npad + 2 * npad + 1)
        Fp = np.eye(npad)
        Fp[:n, :n] = F
        buf[:npad * npad] = Fp.T.ravel()
        buf[npad * npad:npad * npad + n] = est
        if rhs is not None:
            buf[npad * npad + npad:npad * npad + npad + n] = rhs
            buf[-1] = 1.0
        b, pb = _f64(buf)
        self._chk(self.lib.dnagpu_junction_import(self.h, chain, m.h, C.cast(pb, C.c_void_p), n))
        m.n = n

    def junction_payload_get(self, m, n, chain=0):
        """(F, attached vector, rhs or None) of a matrix (dnagpu_junction_export)"""
        npad = ((n + 127) // 128) * 128 if n else 128
        buf = np.zeros(npad * npad + 2 * npad + 1)
        b, pb = _f64(buf)
        self._chk(self.lib.dnagpu_junction_export(self.h, chain, m.h, C.cast(pb, C.c_void_p), b.size))
        F = b[:npad * npad].reshape(npad, npad).T[:n, :n].copy()
        est = b[npad * npad:npad * npad + n].copy()
        rhs = b[npad * npad + npad:npad * npad + npad + n].copy() if b[-1] == 1.0 else None
        return F, est, rhs

    def junction_gather(self, blk_from, src, idx_from, jm, chain=0):
        ix, p = _u32(idx_from)
        self._chk(self.lib.dnagpu_junction_gather(self.h, chain, blk_from, src.h, p, ix.size, jm.h))
        jm.n = 3 * ix.size

    def schur_carry(self, blk, m, idx_out, jm, chain=0):
        """jm <- Schur complement of the other unknowns of m onto the listed stations (dnagpu_schur_carry); m is destroyed.
        Information form (default): junction estimates <- the block's estimates, the reduced right-hand side beside them;
        estimates form (dnagpu_debug_set_info_carry(0)): junction estimates <- estimates + corrections"""
        ix, p = _u32(idx_out)
        self._chk(self.lib.dnagpu_schur_carry(self.h, chain, blk, m.h, p, ix.size, jm.h))
        jm.n = 3 * ix.size

    def block_reduce(self, blk, m, idx_keep, red, keep=None, chain=0):
        """red <- Schur complement of the other unknowns of m onto the listed stations + reduced rhs (dnagpu_block_reduce);
        keep: handle from partial_create, retains the factor for partial_complete"""
        ix, p = _u32(idx_keep)
        self._chk(self.lib.dnagpu_block_reduce(self.h, chain, blk, m.h, p, ix.size, red.h, keep))
        red.n = 3 * ix.size

    def block_form_reduce(self, blk, con_stn, con_w9, idx_keep, red, keep, chain=0):
        """form_normals + add_diag3x3(+1) + block_reduce(keep) in one step, the normals formed directly in the elimination's order"""
        ix, p = _u32(idx_keep)
        cs, pcs = _u32(con_stn)
        cw, pcw = _f64(con_w9)
        self._chk(self.lib.dnagpu_block_form_reduce(self.h, chain, blk, pcs, pcw, cs.size, p, ix.size, red.h, keep))
        red.n = 3 * ix.size

    def partial_create(self, n_max, k_max):
        h = C.c_void_p()
        self._chk(self.lib.dnagpu_partial_create(self.h, n_max, k_max, C.byref(h)))
        return h

    def partial_create_in(self, n_max, k_max, store):
        """the factor's inverse lives in `store` (a matrix created with n_max + 256) between block_reduce and partial_complete"""
        h = C.c_void_p()
        self._chk(self.lib.dnagpu_partial_create_in(self.h, n_max, k_max, store.h, C.byref(h)))
        return h

    def partial_create_spine(self, n_max, k_max, store):
        """light form: the elimination's block factor in `store`, no inverse of the eliminated part until partial_finish"""
        h = C.c_void_p()
        self._chk(self.lib.dnagpu_partial_create_spine(self.h, n_max, k_max, store.h, C.byref(h)))
        return h

    def partial_destroy(self, h):
        self.lib.dnagpu_partial_destroy(self.h, h)

    def partial_complete(self, pf, kk, inv, n, chain=0):
        """inv (order n, natural unknown order) <- inverse of the block whose kept part is now kk"""
        self._chk(self.lib.dnagpu_partial_complete(self.h, chain, pf, kk.h, inv.h))
        inv.n = n

    def partial_complete_factor(self, pf, kk, chain=0):
        self._chk(self.lib.dnagpu_partial_complete_factor(self.h, chain, pf, kk.h))

    def partial_solve(self, blk, pf, chain=0):
        """corrections(blk) <- N^-1 rhs(blk) from the completed factor"""
        self._chk(self.lib.dnagpu_partial_solve(self.h, chain, blk, pf))
        self.sync()

    def partial_finish(self, pf, inv, n, chain=0):
        self._chk(self.lib.dnagpu_partial_finish(self.h, chain, pf, inv.h))
        inv.n = n

    def partial_reduce_rhs(self, blk, pf, red, chain=0):
        self._chk(self.lib.dnagpu_partial_reduce_rhs(self.h, chain, blk, pf, red.h))
        self.sync()

    def junction_scatter(self, dst, idx_to, jm, chain=0):
        ix, p = _u32(idx_to)
        self._chk(self.lib.dnagpu_junction_scatter(self.h, chain, dst.h, p, ix.size, jm.h))

    def junction_rhs(self, blk_to, idx_to, jm, chain=0):
        ix, p = _u32(idx_to)
        self._chk(self.lib.dnagpu_junction_rhs(self.h, chain, blk_to, p, ix.size, jm.h))

    def junction_put_estimates(self, jm, est, chain=0):
        e = np.ascontiguousarray(est, dtype=np.float64)
        self._chk(self.lib.dnagpu_junction_put_estimates(self.h, chain, jm.h, e.ctypes.data_as(c_f64p), e.size // 3))

    def junction_get_estimates(self, jm, chain=0):
        out = np.empty(jm.n, dtype=np.float64)
        self._chk(self.lib.dnagpu_junction_get_estimates(self.h, chain, jm.h, out.ctypes.data_as(c_f64p)))
        return out
