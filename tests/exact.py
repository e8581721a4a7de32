"""An independent solution of a linear (GNSS) network in EXTENDED precision: numpy longdouble (x87 80-bit: 64-bit mantissa, 2 048 times finer
than fp64) from the assembly of the dense design and weight matrices to the Cholesky solve -- no LAPACK, no code shared with the oracle or the
device path beyond the reader of the network files.  What it is for: the reference's own fp64 solver cannot run here, so "within 1e-8 m of the
reference" is argued through the exact answer -- an fp64 solution that is within a few units in the last place of a 4e6 m coordinate of the exact
least-squares solution is within twice that of every other such solution, the reference's included."""
import numpy as np

LD = np.longdouble


def _cholesky_solve(N, B):
    """N X = B for symmetric positive definite N, all in longdouble (column Cholesky with vectorised updates)"""
    n = N.shape[0]
    L = np.array(N, dtype=LD)
    for j in range(n):
        d = np.sqrt(L[j, j])
        L[j:, j] /= d
        if j + 1 < n:
            L[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], L[j + 1:, j])
    L = np.tril(L)
    Y = np.array(B, dtype=LD)
    for j in range(n):                       # forward substitution, column oriented
        Y[j] /= L[j, j]
        if j + 1 < n:
            Y[j + 1:] -= np.multiply.outer(L[j + 1:, j], Y[j]) if Y.ndim > 1 else L[j + 1:, j] * Y[j]
    for j in range(n - 1, -1, -1):           # backward substitution with L^T
        Y[j] /= L[j, j]
        if j:
            Y[:j] -= np.multiply.outer(L[j, :j], Y[j]) if Y.ndim > 1 else L[j, :j] * Y[j]
    return Y


def _inverse(M):
    return _cholesky_solve(np.array(M, dtype=LD), np.eye(M.shape[0], dtype=LD))


def solve(net, fixed_std_dev=1e-6, free_std_dev=10.0, threshold=float(np.float32(0.0005)), max_iterations=10, variances=True):
    """(coordinates, variance matrix, iterations): x <- x + (A' W A + Wc)^-1 A' W (obs - A x), iterated like AdjustSimultaneous (the station
    constraints weight the normals only); CCC / FFF constraints, G baselines and X / Y clusters with their full variance matrices"""
    n, m = 3 * net.n_stations, 3 * net.n_baselines
    A = np.zeros((m, n), dtype=LD)
    for i in range(net.n_baselines):
        for c in range(3):
            if net.stn1[i] != 0xffffffff:
                A[3 * i + c, 3 * int(net.stn1[i]) + c] = -1
            A[3 * i + c, 3 * int(net.stn2[i]) + c] = 1
    W = np.zeros((m, m), dtype=LD)
    voff = 0
    if net.n_clusters == 0:
        # single 'G' baselines: six variance terms each, the upper triangle by columns (xx; xy yy; xz yz zz -- tests/oracle.py Network)
        for i in range(net.n_baselines):
            v = np.array(net.vcv6[6 * i:6 * i + 6], dtype=LD)
            V = np.array([[v[0], v[1], v[3]], [v[1], v[2], v[4]], [v[3], v[4], v[5]]], dtype=LD)
            W[3 * i:3 * i + 3, 3 * i:3 * i + 3] = _inverse(V)
    for c in range(net.n_clusters):
        i0, i1 = int(net.cluster_off[c]), int(net.cluster_off[c + 1])
        nc = 3 * (i1 - i0)
        V = np.array(net.cluster_vcv[voff:voff + nc * nc], dtype=LD).reshape(nc, nc, order="F")
        voff += nc * nc
        W[3 * i0:3 * i1, 3 * i0:3 * i1] = _inverse(V)
    Wc = np.zeros((n, n), dtype=LD)
    for s in range(net.n_stations):
        cst = net.constraints[3 * s:3 * s + 3]
        assert cst in (b"CCC", b"FFF"), "mixed constraints are not restated here"
        sd = LD(fixed_std_dev) if cst == b"CCC" else LD(free_std_dev)
        Wc[3 * s:3 * s + 3, 3 * s:3 * s + 3] = np.eye(3, dtype=LD) / (sd * sd)
    AtW = A.T @ W
    N = AtW @ A + Wc
    x = np.array(net.xyz0, dtype=LD)
    obs = np.array(net.obs, dtype=LD)
    its = 0
    for _ in range(max_iterations):
        its += 1
        dx = _cholesky_solve(N, AtW @ (obs - A @ x))
        x = x + dx
        if float(np.abs(dx).max()) <= threshold:
            break
    return x, (_inverse(N) if variances else None), its


# ---- at size: sparse assembly, blocked factorisation (tools/make_exact_golden.py -> tests/golden/exact_3k.npz) ---------------------------
def _chol_blocked(N, nb=96):
    """lower Cholesky factor of N (longdouble), right-looking by panels of nb columns: the panel by the column algorithm above, the
    trailing update as one matrix product per panel"""
    n = N.shape[0]
    L = np.array(N, dtype=LD)
    for o in range(0, n, nb):
        e = min(n, o + nb)
        for j in range(o, e):
            d = np.sqrt(L[j, j])
            L[j:, j] /= d
            if j + 1 < e:
                L[j + 1:, j + 1:e] -= np.outer(L[j + 1:, j], L[j + 1:e, j])
        if e < n:
            P = L[e:, o:e]
            L[e:, e:] -= P @ P.T
    return np.tril(L)


def _tri_inverse_blocked(L, nb=96):
    """X = L^-1 for lower triangular L (longdouble), block column by block column"""
    n = L.shape[0]
    X = np.zeros((n, n), dtype=LD)
    for o in range(0, n, nb):
        e = min(n, o + nb)
        D = L[o:e, o:e]
        Xd = np.zeros((e - o, e - o), dtype=LD)
        for j in range(e - o):                  # D Xd = I by forward substitution, column by column
            y = np.zeros(e - o, dtype=LD)
            y[j] = LD(1)
            for i in range(j, e - o):
                y[i] = (y[i] - np.dot(D[i, j:i], y[j:i])) / D[i, i]
            Xd[:, j] = y
        X[o:e, o:e] = Xd
    # below the diagonal blocks: X[i, j] = -X[i, i] * sum_{j <= k < i} L[i, k] X[k, j], block row by block row
    for o in range(0, n, nb):
        e = min(n, o + nb)
        if o:
            X[o:e, :o] = -(X[o:e, o:e] @ (L[o:e, :o] @ X[:o, :o]))
    return X


def solve_sparse(net, fixed_std_dev=1e-6, free_std_dev=10.0, threshold=float(np.float32(0.0005)), max_iterations=10):
    """the same solution as solve() for networks of thousands of unknowns: the normals are assembled block by block from the baselines (no
    dense design matrix), factored by panels.  Single 'G' baselines and CCC / FFF constraints.  Iterated like AdjustSimultaneous (the station
    constraints weight the normals only, so the iteration count is part of the answer).  Returns (x, N^-1, iterations) in longdouble."""
    assert net.n_clusters == 0
    n = 3 * net.n_stations
    N = np.zeros((n, n), dtype=LD)
    Wb = []
    for i in range(net.n_baselines):
        v = np.array(net.vcv6[6 * i:6 * i + 6], dtype=LD)
        V = np.array([[v[0], v[1], v[3]], [v[1], v[2], v[4]], [v[3], v[4], v[5]]], dtype=LD)
        W = _inverse(V)
        Wb.append(W)
        a, b = 3 * int(net.stn1[i]), 3 * int(net.stn2[i])
        N[a:a + 3, a:a + 3] += W
        N[b:b + 3, b:b + 3] += W
        N[a:a + 3, b:b + 3] -= W
        N[b:b + 3, a:a + 3] -= W
    for s in range(net.n_stations):
        cst = net.constraints[3 * s:3 * s + 3]
        assert cst in (b"CCC", b"FFF")
        sd = LD(fixed_std_dev) if cst == b"CCC" else LD(free_std_dev)
        N[3 * s:3 * s + 3, 3 * s:3 * s + 3] += np.eye(3, dtype=LD) / (sd * sd)
    L = _chol_blocked(N)
    X = _tri_inverse_blocked(L)
    Ninv = X.T @ X
    x = np.array(net.xyz0, dtype=LD)
    obs = np.array(net.obs, dtype=LD).reshape(-1, 3)
    its = 0
    for _ in range(max_iterations):
        its += 1
        r = np.zeros(n, dtype=LD)
        xs = x.reshape(-1, 3)
        for i in range(net.n_baselines):
            a, b = int(net.stn1[i]), int(net.stn2[i])
            wb = Wb[i] @ (obs[i] - (xs[b] - xs[a]))
            r[3 * b:3 * b + 3] += wb
            r[3 * a:3 * a + 3] -= wb
        dx = Ninv @ r
        x = x + dx
        if float(np.abs(dx).max()) <= threshold:
            break
    return x, Ninv, its
