"""A dense numpy implementation of the block-step interface of tests/parallel_harness.py, used to exercise the
multi-rank schedule and its messaging on CPU (gloo).  Test infrastructure only: it restates the same steps as
dna_adjust::Phased{Forward,Reverse,Combine,Finalise}Block with numpy.linalg on small networks."""
import numpy as np
import torch


class NumpyBlockBackend:
    def __init__(self, net, fixed_std_dev=1e-6, free_std_dev=10.0, threshold=float(np.float32(0.0005)), max_iterations=10, condensed=True):
        self.comm_device = torch.device("cpu")
        self.net = net
        self.n_blocks = B = net.n_blocks
        self.threshold = threshold
        self.max_iterations = max_iterations
        self.blk = []
        seen_f, order = set(), []
        for k in range(B):
            isl = net.isl[net.isl_off[k]:net.isl_off[k + 1]]
            jsl = net.jsl[net.jsl_off[k]:net.jsl_off[k + 1]]
            st = np.array(sorted(set(int(s) for s in isl) | set(int(s) for s in jsl)), dtype=np.int64)
            loc = {int(s): i for i, s in enumerate(st)}
            cml = net.cml[net.cml_off[k]:net.cml_off[k + 1]]
            d = dict(st=st, loc=loc, jsl=[int(s) for s in jsl], cml=cml,
                     s1=np.array([loc[int(net.stn1[i])] for i in cml], dtype=np.int64),
                     s2=np.array([loc[int(net.stn2[i])] for i in cml], dtype=np.int64))
            d["first_fwd"] = [int(s) not in seen_f for s in st]
            seen_f |= set(int(s) for s in st)
            x = np.concatenate([net.xyz0[3 * s:3 * s + 3] for s in st])
            d["orig"], d["est"], d["rig"] = x.copy(), x.copy(), x.copy()
            d["rigvar"] = None
            self.blk.append(d)
        seen_r = set()
        for k in range(B - 1, -1, -1):
            d = self.blk[k]
            d["first_rev"] = [int(s) not in seen_r for s in d["st"]]
            seen_r |= set(int(s) for s in d["st"])
        self.flags_ = []
        for k in range(B):
            first = k == 0 or net.net_id[k] != net.net_id[k - 1]
            last = k == B - 1 or net.net_id[k] != net.net_id[k + 1]
            self.flags_.append((first, last, first and last))
        self.wc, self.wf = 1.0 / fixed_std_dev ** 2, 1.0 / free_std_dev ** 2
        self.W = []
        for i in range(net.n_baselines):
            v = net.vcv6[6 * i:6 * i + 6]
            V = np.array([[v[0], v[1], v[3]], [v[1], v[2], v[4]], [v[3], v[4], v[5]]])
            self.W.append(np.linalg.inv(V))
        for k in range(B):
            self._compute_b(k)
        self.jfwd, self.jrev, self.red = {}, {}, {}
        self.condensed_ = condensed
        self.maxcorr = 0.0
        self.iteration = 0
        self.history = []

    # ---- helpers ----
    def _weight(self, s):
        c = self.net.constraints[3 * s:3 * s + 3]
        return self.wc if c == b"CCC" else self.wf

    def _compute_b(self, k):
        d = self.blk[k]
        x = d["est"].reshape(-1, 3)
        obs = np.stack([self.net.obs[3 * i:3 * i + 3] for i in d["cml"]]) if len(d["cml"]) else np.zeros((0, 3))
        d["b"] = obs - (x[d["s2"]] - x[d["s1"]])

    def _normals(self, k):
        d = self.blk[k]
        n = 3 * len(d["st"])
        N = np.zeros((n, n))
        rhs = np.zeros(n)
        for j, i in enumerate(d["cml"]):
            a, b = 3 * d["s1"][j], 3 * d["s2"][j]
            W = self.W[i]
            N[a:a + 3, a:a + 3] += W
            N[b:b + 3, b:b + 3] += W
            N[a:a + 3, b:b + 3] -= W
            N[b:b + 3, a:a + 3] -= W
            wb = W @ d["b"][j]
            rhs[a:a + 3] -= wb
            rhs[b:b + 3] += wb
        return N, rhs

    def _constraints(self, k, N, which, sign=1.0):
        d = self.blk[k]
        for p, s in enumerate(d["st"]):
            if which == "fwd" and not d["first_fwd"][p]:
                continue
            if which == "rev" and not d["first_rev"][p]:
                continue
            if which == "cmb" and d["first_fwd"][p]:
                continue
            N[3 * p:3 * p + 3, 3 * p:3 * p + 3] += sign * self._weight(int(s)) * np.eye(3)

    def _rows(self, k, stations):
        loc = self.blk[k]["loc"]
        return np.array([3 * loc[s] + c for s in stations for c in range(3)], dtype=np.int64)

    def _junction_in(self, k, N, rhs, stations, payload):
        if not stations:
            return
        WJ, est = payload
        r = self._rows(k, stations)
        N[np.ix_(r, r)] += WJ
        rhs[r] += WJ @ (est - self.blk[k]["est"][r])

    def _solve(self, k, N, rhs):
        Ninv = np.linalg.inv(N)
        corr = Ninv @ rhs
        d = self.blk[k]
        d["est"] = d["est"] + corr
        i = int(np.argmax(np.abs(corr)))
        self._Ninv = Ninv
        return float(corr[i])

    # ---- interface of parallel.run_phased ----
    def flags(self, k):
        return self.flags_[k]

    def n_stations(self, k):
        return len(self.blk[k]["st"])

    def begin_iteration(self):
        self.maxcorr = 0.0
        self.iteration += 1

    def note_correction(self, mv):
        if abs(mv) > abs(self.maxcorr):
            self.maxcorr = mv

    def max_correction(self):
        return self.maxcorr

    def forward_block(self, k):
        f, l, i = self.flags_[k]
        d = self.blk[k]
        N, rhs = self._normals(k)
        self._constraints(k, N, "fwd")
        if not f and not i:
            self._junction_in(k, N, rhs, self.blk[k - 1]["jsl"], self.jfwd[k - 1])
        mv = self._solve(k, N, rhs)
        if l or i:
            self.note_correction(mv)
            d["rig"] = d["est"].copy()
            d["rigvar"] = self._Ninv
            return mv
        if d["jsl"]:
            r = self._rows(k, d["jsl"])
            self.jfwd[k] = (np.linalg.inv(self._Ninv[np.ix_(r, r)]), d["est"][r].copy())
        return mv

    def reverse_block(self, k):
        f, l, i = self.flags_[k]
        d = self.blk[k]
        d["est"] = d["orig"].copy()
        N, rhs = self._normals(k)
        if not l:
            self._junction_in(k, N, rhs, d["jsl"], self.jrev[k])
        self._constraints(k, N, "rev")
        mv = self._solve(k, N, rhs)
        if not f and self.blk[k - 1]["jsl"]:
            r = self._rows(k, self.blk[k - 1]["jsl"])
            self.jrev[k - 1] = (np.linalg.inv(self._Ninv[np.ix_(r, r)]), d["est"][r].copy())
        return mv

    def combine_block(self, k):
        d = self.blk[k]
        d["est"] = d["orig"].copy()
        N, rhs = self._normals(k)
        self._junction_in(k, N, rhs, d["jsl"], self.jrev[k])
        self._constraints(k, N, "rev")
        self._junction_in(k, N, rhs, self.blk[k - 1]["jsl"], self.jfwd[k - 1])
        self._constraints(k, N, "cmb", -1.0)
        return self._solve(k, N, rhs)

    def finalise_block(self, k):
        d = self.blk[k]
        d["rig"] = d["est"].copy()
        d["rigvar"] = self._Ninv
        d["orig"] = d["rig"].copy()

    def end_iteration(self):
        self.history.append(self.maxcorr)
        iterate = abs(self.maxcorr) > self.threshold and self.iteration < self.max_iterations
        if iterate:
            for k in range(self.n_blocks):
                d = self.blk[k]
                if self.flags_[k][1]:
                    d["orig"] = d["rig"].copy()
                d["est"] = d["rig"].copy()
                self._compute_b(k)
        return iterate

    def finish(self):
        return 1 if (self.iteration == self.max_iterations and abs(self.maxcorr) > self.threshold) else 0

    def _njs(self, k):
        return 3 * len(self.blk[k]["jsl"])

    def junction_tensor(self, kind, k):
        n = self._njs(k)
        return torch.empty(n * n + n, dtype=torch.float64)

    def export_junction(self, kind, k):
        WJ, est = (self.jfwd if kind == 0 else self.jrev)[k]
        return torch.from_numpy(np.concatenate([WJ.ravel(), est]))

    def import_junction(self, kind, k, t):
        n = self._njs(k)
        a = t.numpy().copy()
        (self.jfwd if kind == 0 else self.jrev)[k] = (a[:n * n].reshape(n, n), a[n * n:])

    # ---- the condensed schedule (parallel.run_phased_condensed), dense numpy restatement -------------------------------
    def condensed(self):
        return self.condensed_

    def _kept(self, k):
        """stations block k shares with its neighbours, in block order; positions of JSL(k-1) / JSL(k) among them"""
        f, l, i = self.flags_[k]
        d = self.blk[k]
        prev = [] if (f or i) else self.blk[k - 1]["jsl"]
        nxt = [] if (l or i) else d["jsl"]
        keep = sorted(set(d["loc"][s] for s in prev) | set(d["loc"][s] for s in nxt))
        pos = {p: q for q, p in enumerate(keep)}
        return keep, [pos[d["loc"][s]] for s in prev], [pos[d["loc"][s]] for s in nxt]

    @staticmethod
    def _unk(stations):
        return np.array([3 * p + c for p in stations for c in range(3)], dtype=np.int64)

    def condense_block(self, k):
        d = self.blk[k]
        keep, _, _ = self._kept(k)
        if not keep:
            return
        N, rhs = self._normals(k)
        inner = [p for p in range(len(d["st"])) if p not in set(keep)]
        for p in inner:      # a station of one block only appears first in it in both directions
            assert d["first_fwd"][p] and d["first_rev"][p]
            N[3 * p:3 * p + 3, 3 * p:3 * p + 3] += self._weight(int(d["st"][p])) * np.eye(3)
        rk, ri = self._unk(keep), self._unk(inner)
        if len(ri):
            X = np.linalg.solve(N[np.ix_(ri, ri)], np.column_stack([N[np.ix_(ri, rk)], rhs[ri]]))
            S = N[np.ix_(rk, rk)] - N[np.ix_(rk, ri)] @ X[:, :-1]
            r = rhs[rk] - N[np.ix_(rk, ri)] @ X[:, -1]
        else:
            S, r = N[np.ix_(rk, rk)], rhs[rk]
        self.red[k] = (S, r)

    def _condensed_step(self, k, direction):
        """one chain step on the condensed block: returns (weights, estimates) of the stations carried on"""
        f, l, i = self.flags_[k]
        d = self.blk[k]
        keep, cprev, cnext = self._kept(k)
        S, r = self.red[k]
        S, r = S.copy(), r.copy()
        x0 = d["orig"][self._unk(keep)]
        which = "first_fwd" if direction == "fwd" else "first_rev"
        for q, p in enumerate(keep):
            if d[which][p]:
                S[3 * q:3 * q + 3, 3 * q:3 * q + 3] += self._weight(int(d["st"][p])) * np.eye(3)
        cin, payload, cout = (cprev, self.jfwd.get(k - 1), cnext) if direction == "fwd" else (cnext, self.jrev.get(k), cprev)
        if cin and payload is not None:
            WJ, est = payload
            rr = self._unk(cin)
            S[np.ix_(rr, rr)] += WJ
            r[rr] += WJ @ (est - x0[rr])
        ro = self._unk(cout)
        re = self._unk([q for q in range(len(keep)) if q not in set(cout)])
        if len(re):
            X = np.linalg.solve(S[np.ix_(re, re)], np.column_stack([S[np.ix_(re, ro)], r[re]]))
            W = S[np.ix_(ro, ro)] - S[np.ix_(ro, re)] @ X[:, :-1]
            ro_rhs = r[ro] - S[np.ix_(ro, re)] @ X[:, -1]
        else:
            W, ro_rhs = S[np.ix_(ro, ro)], r[ro]
        return W, x0[ro] + np.linalg.solve(W, ro_rhs)

    def condensed_forward(self, k):
        f, l, i = self.flags_[k]
        if i or l or not self.blk[k]["jsl"] or self.flags_[k + 1][2]:
            return
        self.jfwd[k] = self._condensed_step(k, "fwd")

    def condensed_reverse(self, k):
        f, l, i = self.flags_[k]
        if i or f or not self.blk[k - 1]["jsl"]:
            return
        self.jrev[k - 1] = self._condensed_step(k, "rev")

    def rigorous_block(self, k):
        f, l, i = self.flags_[k]
        if l or i:
            return self.forward_block(k)
        mv = self.reverse_block(k) if f else self.combine_block(k)
        self.note_correction(mv)
        self.finalise_block(k)
        return mv

    def condense_blocks(self, blocks):
        for k in blocks:
            self.condense_block(k)

    def condensed_chains(self):
        for k in range(self.n_blocks):
            self.condensed_forward(k)
        for k in range(self.n_blocks - 1, -1, -1):
            self.condensed_reverse(k)

    def rigorous_blocks(self, blocks):
        for k in blocks:
            self.rigorous_block(k)

    def condensed_tensor(self, k):
        n = 3 * len(self._kept(k)[0])
        return torch.empty(n * n + n, dtype=torch.float64) if n else None

    def export_condensed(self, k):
        if k not in self.red:
            return None
        S, r = self.red[k]
        return torch.from_numpy(np.concatenate([S.ravel(), r]))

    def import_condensed(self, k, t):
        n = 3 * len(self._kept(k)[0])
        a = t.numpy().copy()
        self.red[k] = (a[:n * n].reshape(n, n), a[n * n:])

    def get_coords(self, k):
        return self.blk[k]["rig"].copy()

    def set_coords(self, k, xyz):
        d = self.blk[k]
        d["orig"], d["est"], d["rig"] = xyz.copy(), xyz.copy(), xyz.copy()


# ---- two-level condensed chains (dna_adjust_dist.cpp: ReduceOwnRun / ExchangeRuns / ScanRuns / OwnRunChains), dense numpy restatement ----
from tests.partition import contiguous_owners      # noqa: E402,F401  (kept importable from here)


class TwoLevelMixin:
    """added to NumpyBlockBackend below: a run of condensed blocks reduced to the stations of its two ends, the chains over the runs,
    the chains inside a run"""

    def _gid(self, k, keep_pos_list):
        return [int(self.blk[k]["st"][p]) for p in keep_pos_list]

    def run_ends(self, a, b):
        """global ids of the junction stations towards the previous run (JSL(a-1) order) and the next run (JSL(b) order)"""
        L = [] if self.flags_[a][0] else list(self.blk[a - 1]["jsl"])
        R = [] if self.flags_[b][1] else list(self.blk[b]["jsl"])
        return L, R

    def _system_of_block(self, k):
        keep, _, _ = self._kept(k)
        S, r = self.red[k]
        return self._gid(k, keep), S.copy(), r.copy()

    @staticmethod
    def _merge(sys1, sys2):
        g1, S1, r1 = sys1
        g2, S2, r2 = sys2
        U = sorted(set(g1) | set(g2))
        pos = {g: i for i, g in enumerate(U)}
        n = 3 * len(U)
        S, r = np.zeros((n, n)), np.zeros(n)
        for g, Sx, rx in ((g1, S1, r1), (g2, S2, r2)):
            idx = np.array([3 * pos[s] + c for s in g for c in range(3)], dtype=np.int64)
            S[np.ix_(idx, idx)] += Sx
            r[idx] += rx
        return U, S, r

    @staticmethod
    def _eliminate(system, stay):
        g, S, r = system
        stay = [s for s in g if s in set(stay)]
        rk = np.array([3 * g.index(s) + c for s in stay for c in range(3)], dtype=np.int64)
        ri = np.array([3 * i + c for i, s in enumerate(g) if s not in set(stay) for c in range(3)], dtype=np.int64)
        if len(ri) == 0:
            return stay, S[np.ix_(rk, rk)], r[rk]
        X = np.linalg.solve(S[np.ix_(ri, ri)], np.column_stack([S[np.ix_(ri, rk)], r[ri]]))
        return stay, S[np.ix_(rk, rk)] - S[np.ix_(rk, ri)] @ X[:, :-1], r[rk] - S[np.ix_(rk, ri)] @ X[:, -1]

    def _add_first_constraints(self, system, blocks, which, only=None, exclude=None):
        """direction dependent constraint of every station of `system` whose first appearance (forward / reverse) is in one of `blocks`"""
        g, S, r = system
        for k in blocks:
            d = self.blk[k]
            for p, s in enumerate(d["st"]):
                s = int(s)
                if s not in g or not d[which][p]:
                    continue
                if only is not None and s not in only:
                    continue
                if exclude is not None and s in exclude:
                    continue
                q = 3 * g.index(s)
                S[q:q + 3, q:q + 3] += self._weight(s) * np.eye(3)

    def reduce_run(self, a, b):
        """level 1: (stations, S, r) of the run a..b on the stations of its two ends"""
        L, R = self.run_ends(a, b)
        ends = set(L) | set(R)
        system = self._system_of_block(a)
        for k in range(a + 1, b + 1):
            system = self._merge(system, self._system_of_block(k))
            # stations that leave inside the run take their constraint where the forward chain adds it (first appearance)
            self._add_first_constraints(system, [a, k] if k == a + 1 else [k], "first_fwd", exclude=ends)
            nxt = [] if self.flags_[k][1] else list(self.blk[k]["jsl"])
            system = self._eliminate(system, set(L) | set(nxt))
        g, S, r = system
        assert set(g) == ends, (sorted(g), sorted(ends))
        return system

    def _x0(self, stations, a, b):
        out = np.zeros(3 * len(stations))
        for i, s in enumerate(stations):
            k = a if s in self.blk[a]["loc"] else b
            p = self.blk[k]["loc"][s]
            out[3 * i:3 * i + 3] = self.blk[k]["orig"][3 * p:3 * p + 3]
        return out

    def scan_runs(self, runs, systems):
        """level 2: forward and reverse chain over the runs; leaves jfwd at the last block of every run but the last and jrev at
        the block before every run but the first"""
        W = len(runs)
        for direction in ("fwd", "rev"):
            order = range(0, W - 1) if direction == "fwd" else range(W - 1, 0, -1)
            for r in order:
                a, b = runs[r]
                L, R = self.run_ends(a, b)
                g, S, rr = systems[r]
                g, S, rr = list(g), S.copy(), rr.copy()
                x0 = self._x0(g, a, b)
                self._add_first_constraints((g, S, rr), range(a, b + 1), "first_fwd" if direction == "fwd" else "first_rev")
                cin, payload, cout = (L, self.jfwd.get(a - 1), R) if direction == "fwd" else (R, self.jrev.get(b), L)
                if cin and payload is not None:
                    WJ, est = payload
                    idx = np.array([3 * g.index(s) + c for s in cin for c in range(3)], dtype=np.int64)
                    S[np.ix_(idx, idx)] += WJ
                    rr[idx] += WJ @ (est - x0[idx])
                stay, Wm, rhs = self._eliminate((g, S, rr), cout)
                # (junction list order, as the block-level chains keep their matrices)
                perm = np.array([3 * stay.index(s) + c for s in cout for c in range(3)], dtype=np.int64)
                Wm, rhs = Wm[np.ix_(perm, perm)], rhs[perm]
                x0o = self._x0(cout, a, b)
                out = (Wm, x0o + np.linalg.solve(Wm, rhs))
                if direction == "fwd":
                    self.jfwd[b] = out
                else:
                    self.jrev[a - 1] = out

    def own_chains(self, a, b):
        for k in range(a, b):
            self.condensed_forward(k)
        for k in range(b, a, -1):
            self.condensed_reverse(k)


for _name in ("_gid", "run_ends", "_system_of_block", "_merge", "_eliminate", "_add_first_constraints", "reduce_run", "_x0", "scan_runs",
              "own_chains"):
    setattr(NumpyBlockBackend, _name, TwoLevelMixin.__dict__[_name])


def run_two_level(be, dist, rank, world, max_iterations=10):
    """the condensed schedule with two-level chains across `world` gloo ranks (the message pattern of dna_adjust_dist.cpp:
    one broadcast per RANK of the run's system, coordinates by all_reduce); returns (status, iterations, corrections, owners)"""
    from tests import parallel_harness as parallel
    B = be.n_blocks
    owner = contiguous_owners([float(be.n_stations(k)) ** 3 for k in range(B)], world)
    runs = [(owner.index(r), B - 1 - owner[::-1].index(r)) for r in range(world)]
    offs = np.zeros(B + 1, dtype=np.int64)
    for k in range(B):
        offs[k + 1] = offs[k] + 3 * be.n_stations(k)
    corrections = []
    for _ in range(max_iterations):
        be.begin_iteration()
        a, b = runs[rank]
        mine = list(range(a, b + 1))
        be.condense_blocks(mine)
        own = be.reduce_run(a, b)                                          # level 1
        systems, pending = [], []
        for r, (ra, rb) in enumerate(runs):                                # exchange: one system per rank
            L, R = be.run_ends(ra, rb)
            g = sorted(set(L) | set(R))
            n = 3 * len(g)
            t = torch.from_numpy(np.concatenate([own[1].ravel(), own[2]])) if r == rank else torch.empty(n * n + n, dtype=torch.float64)
            assert r != rank or own[0] == g
            pending.append((g, n, t, dist.broadcast(t, src=r, async_op=True)))
        for g, n, t, w in pending:
            w.wait()
            arr = t.numpy()
            systems.append((g, arr[:n * n].reshape(n, n).copy(), arr[n * n:].copy()))
        be.scan_runs(runs, systems)                                        # level 2 (every rank)
        be.own_chains(a, b)                                                # level 3
        be.rigorous_blocks(mine)
        parallel._sync_coordinates(be, dist, rank, world, lambda k: owner[k], offs, be.comm_device)
        corrections.append(be.max_correction())
        if not be.end_iteration():
            break
    return be.finish(), len(corrections), corrections, owner
