"""GPU smoke/perf check of the dense layer (run through gpurun)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynadjust_amd.device import DeviceContext, pack_lower, unpack_lower

def spd(n, rng, cond_spike=False):
    A = rng.standard_normal((n, n + 8))
    M = A @ A.T / n + np.eye(n) * 0.5
    if cond_spike:
        d = np.ones(n); d[::7] = 1e6
        M = M * np.outer(d, d) ** 0.5
    return (M + M.T) * 0.5

ctx = DeviceContext(0)
rng = np.random.default_rng(1)
ok = True
for n in [2, 3, 5, 64, 128, 129, 255, 256, 257, 450, 1023, 1500, 3000]:
    M = spd(n, rng)
    ap = pack_lower(M)
    inv = unpack_lower(ctx.cholesky_inverse_packed(ap, n), n)
    ref = np.linalg.inv(M)
    err = np.abs(inv - ref).max() / np.abs(ref).max()
    inv2 = unpack_lower(ctx.cholesky_inverse_packed(ap, n, True), n)
    err2 = np.abs(inv2 - ref).max() / np.abs(ref).max()
    x = rng.standard_normal(n)
    y = ctx.multiply_sym_packed(ap, x, n)
    err3 = np.abs(y - M @ x).max() / np.abs(M @ x).max()
    print(f"n={n:5d} inv relerr {err:.2e} scaled {err2:.2e} symv {err3:.2e}", flush=True)
    ok &= err < 1e-9 and err2 < 1e-9 and err3 < 1e-12
# non positive definite
M = spd(300, rng); M[150, 150] = -1.0
try:
    ctx.cholesky_inverse_packed(pack_lower(M), 300)
    print("ERROR: indefinite matrix accepted"); ok = False
except Exception as e:
    print("indefinite ->", e, "info", ctx.last_info())
print("CORRECT" if ok else "WRONG", flush=True)

# performance of the in-place inverse
ctx.profile_enable(True)
for n in [4096, 8192, 16384, 30000]:
    m = ctx.matrix(n)
    m.reset(n)
    # diagonally dominant SPD without host O(n^2) work: N = I*4 (+ nothing) is too easy on
    # DVFS; use a banded random SPD built from a small packed upload when n is moderate
    if n <= 8192:
        M = spd(n, rng)
        m.upload_packed(pack_lower(M), n)
    else:
        ns = n // 3
        ctx.add_diag3x3(m, np.arange(ns, dtype=np.uint32), np.tile(np.array([4.0, 1, .5, 1, 5, .25, .5, .25, 6]), ns))
        if n - 3 * ns:
            m.reset(3 * ns)
            ctx.add_diag3x3(m, np.arange(ns, dtype=np.uint32), np.tile(np.array([4.0, 1, .5, 1, 5, .25, .5, .25, 6]), ns))
            n = 3 * ns
    ctx.sync()
    if n > 8192:
        keep = ctx.matrix(n); ctx.lib.dnagpu_matrix_copy(ctx.h, 0, keep.h, m.h)
        m.invert()   # warm-up (plans, clocks)
        ctx.lib.dnagpu_matrix_copy(ctx.h, 0, m.h, keep.h); keep.close()
    ctx.sync()
    ctx.profile_reset()
    t0 = time.perf_counter()
    m.invert()
    ctx.sync()
    dt = time.perf_counter() - t0
    p = ctx.profile_get()
    print(f"n={n}: invert {dt*1e3:.1f} ms  ref-equivalent n^3 = {n**3/dt/1e12:.2f} TFLOP/s | gemm launches {p['launches']} "
          f"flops {p['gemm_flops']:.3e} gemm time {p['gemm_ms']:.1f} ms -> {p['gemm_flops']/p['gemm_ms']/1e9:.2f} TFLOP/s", flush=True)
    if n <= 8192:
        inv = unpack_lower(m.download_packed(), n)
        r = np.abs(inv @ M - np.eye(n)).max()
        print(f"   residual |inv*M - I|max = {r:.2e}")
    m.close()
ctx.close()
