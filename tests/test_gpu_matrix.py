"""L2 seam on the device (dnagpu_cholesky_inverse_packed / dnagpu_multiply_sym_packed) against the
reference's known-answer vectors, the LAPACK golden fixtures and the CPU oracle."""
import json
import os

import numpy as np
import pytest

from dynadjust_amd._lib import DnaGpuError
from dynadjust_amd.device import pack_lower, unpack_lower

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "matrix_golden.json")))


@pytest.fixture(scope="module")
def lap(golden_dir):
    return np.load(os.path.join(golden_dir, "lapack_golden.npz"))


def test_reference_known_answers(gpu_ctx, kat):
    c = kat["cholesky_inverse_3x3"]
    M = np.array(c["matrix"])
    inv = unpack_lower(gpu_ctx.cholesky_inverse_packed(pack_lower(M), 3), 3)
    assert np.abs(inv - np.array(c["inverse"])).max() < c["tol"]
    for key in ("indefinite_2x2", "singular_2x2"):
        with pytest.raises(DnaGpuError) as e:
            gpu_ctx.cholesky_inverse_packed(pack_lower(np.array(kat[key]["matrix"])), 2)
        assert e.value.code == -4 and "Matrix inversion failed, the matrix is singular." in str(e.value)
    c = kat["multiply_sym_4x4"]
    A = np.array(c["matrix"], float)
    for col in range(2):
        y = gpu_ctx.multiply_sym_packed(pack_lower(A), np.array(c["rhs"])[:, col], 4)
        assert np.abs(y - np.array(c["product"])[:, col]).max() < c["tol"]
    c = kat["packed_end_to_end_3x3"]
    M = np.array(c["matrix"])
    inv = gpu_ctx.cholesky_inverse_packed(pack_lower(M), 3)
    y = gpu_ctx.multiply_sym_packed(inv, np.array(c["rhs"]), 3)
    assert np.abs(y - np.linalg.solve(M, np.array(c["rhs"]))).max() < 1e-11
    c = kat["packed_5x5"]
    M = np.array(c["matrix"], float)
    inv = unpack_lower(gpu_ctx.cholesky_inverse_packed(pack_lower(M), 5), 5)
    assert np.abs(inv @ M - np.eye(5)).max() < 1e-12


@pytest.mark.parametrize("n", [3, 6, 129, 300])
def test_lapack_golden(gpu_ctx, lap, n):
    ap = lap[f"ap_{n}"]
    ref = lap[f"inv_{n}"]
    inv = gpu_ctx.cholesky_inverse_packed(ap, n)
    assert np.abs(inv - ref).max() / np.abs(ref).max() < 1e-9          # cond ~1e12-1e14
    refs = lap[f"inv_scaled_{n}"]
    inv_s = gpu_ctx.cholesky_inverse_packed(ap, n, True)
    assert np.abs(inv_s - refs).max() / np.abs(refs).max() < 1e-11     # scale_normals_to_unity path
    y = gpu_ctx.multiply_sym_packed(ap, lap[f"x_{n}"], n)
    assert np.abs(y - lap[f"Ax_{n}"]).max() / np.abs(lap[f"Ax_{n}"]).max() < 1e-13


@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 255, 256, 257, 640, 1000])
def test_edge_sizes_against_oracle(gpu_ctx, orc, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n + 3))
    M = A @ A.T / n + np.eye(n) * 0.5
    ap = pack_lower(M)
    ref, info = orc.cholesky_inverse_packed(ap, n)
    assert info == 0
    inv = gpu_ctx.cholesky_inverse_packed(ap, n)
    assert np.abs(inv - ref).max() / np.abs(ref).max() < 1e-11
    x = rng.standard_normal(n)
    assert np.abs(gpu_ctx.multiply_sym_packed(ap, x, n) - orc.multiply_sym_packed(ap, x, n)).max() < 1e-11 * max(1.0, np.abs(M @ x).max())


def _dense_spd(n, seed):
    """dense, well-conditioned-but-not-trivial SPD matrix: a random Gram matrix of rank n/8 (every off-diagonal panel full)
    plus a diagonal that spans three decades"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, max(8, n // 8)))
    M = A @ A.T / A.shape[1]
    M[np.diag_indices(n)] += 10.0 ** rng.uniform(-1.0, 2.0, n)
    return M


@pytest.fixture
def big_tiles_only(built):
    """every GEMM launch of the inverse goes through the 128-tile throughput kernel (gemm_f64_dma_kernel), whatever its size"""
    old = built.dnagpu_debug_set_small_tiles(0)
    yield
    built.dnagpu_debug_set_small_tiles(old)


@pytest.mark.parametrize("n", [129, 255, 257, 640, 1000, 1500])
def test_throughput_kernel_against_oracle_at_small_orders(gpu_ctx, orc, big_tiles_only, n):
    """dnagpu_debug_set_small_tiles(0): the LDS-DMA kernel -- swizzled S layout, descending k walk, tile tables -- runs the launches the
    64-tile kernel normally takes, at orders where the scalar oracle answers at once; element-wise comparison"""
    M = _dense_spd(n, n)
    ap = pack_lower(M)
    ref, info = orc.cholesky_inverse_packed(ap, n)
    assert info == 0
    inv = gpu_ctx.cholesky_inverse_packed(ap, n)
    assert np.abs(inv - ref).max() / np.abs(ref).max() < 1e-11
    inv_s = gpu_ctx.cholesky_inverse_packed(ap, n, True)
    assert np.abs(inv_s - ref).max() / np.abs(ref).max() < 1e-11


@pytest.mark.parametrize("n", [100, 257, 640, 1000, 1500, 2304, 4096])
def test_tiny_launches_on_32_tiles_have_the_bits_of_64_tiles(gpu_ctx, built, orc, n):
    """dnagpu_debug_set_tiny_tiles (round 4): products of fewer than 64 128-tiles run on 32 x 32 block tiles -- sixteen times the workgroups of the 128-tile
    shape, so that the bottom of the recursion and the chains on condensed blocks occupy more than a handful of CUs.  An element's k order does
    not depend on the tile it is computed in: the inverse is bit for bit the one with the 32-tile shape off, and within 1e-11 of the oracle."""
    M = _dense_spd(n, n + 7)
    ap = pack_lower(M)
    inv32 = gpu_ctx.cholesky_inverse_packed(ap, n)
    old = built.dnagpu_debug_set_tiny_tiles(0)
    try:
        inv64 = gpu_ctx.cholesky_inverse_packed(ap, n)
    finally:
        built.dnagpu_debug_set_tiny_tiles(old)
    assert old == 64
    assert np.array_equal(inv32, inv64)
    if n <= 1500:            # (the scalar oracle answers at once up to here; beyond, the bits of the 64-tile shape are the claim)
        ref, info = orc.cholesky_inverse_packed(ap, n)
        assert info == 0
        assert np.abs(inv32 - ref).max() / np.abs(ref).max() < 1e-11


@pytest.mark.parametrize("n", [2304, 4096, 6016])
def test_dense_inverse_against_oracle_beyond_the_small_launch_threshold(gpu_ctx, orc, n):
    """n >= 2 304 (T = 18: 171 lower tiles > SMALL_LAUNCH_TILES = 160): the top-level launches of the recursion run on the
    128-tile kernel in its normal configuration.  Dense SPD input, element-wise against the oracle (LAPACK = the MKL runtime
    the reference links; the oracle's built-in Cholesky where MKL is absent -- a minute at n = 6 016)"""
    M = _dense_spd(n, n)
    ap = pack_lower(M)
    have_mkl = orc.use_mkl(True)
    try:
        ref, info = orc.cholesky_inverse_packed(ap, n)
    finally:
        orc.use_mkl(False)
    assert info == 0
    inv = gpu_ctx.cholesky_inverse_packed(ap, n)
    err = np.abs(inv - ref).max() / np.abs(ref).max()
    assert err < 1e-11, (err, have_mkl)
    # and against the definition, independent of any LAPACK: N^-1 N = I on probe vectors
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n)
    y = unpack_lower(inv, n) @ (M @ x)
    assert np.abs(y - x).max() < 1e-10


def test_failed_allocation_inside_an_inverse_is_reported(built):
    """a tile-table allocation failing in the middle of the recursion must come back as DNAGPU_ENOMEM (never DNAGPU_OK with a
    skipped launch), and the context must work again afterwards"""
    from dynadjust_amd.device import DeviceContext
    n = 700
    M = _dense_spd(n, 5)
    ap = pack_lower(M)
    for nth in (1, 3, 7):
        ctx = DeviceContext(0)                   # fresh context: empty table cache, so the nth table allocation does happen
        try:
            built.dnagpu_debug_fail_allocation(nth)
            with pytest.raises(DnaGpuError) as e:
                ctx.cholesky_inverse_packed(ap, n)
            assert e.value.code == -2 and "allocation" in str(e.value)
            built.dnagpu_debug_fail_allocation(0)
            # the same context, afterwards: the tables that were missing are built now
            inv = unpack_lower(ctx.cholesky_inverse_packed(ap, n), n)
            assert np.abs(inv @ M - np.eye(n)).max() < 1e-10
        finally:
            built.dnagpu_debug_fail_allocation(0)
            ctx.close()


def test_empty_matrix(gpu_ctx):
    assert gpu_ctx.cholesky_inverse_packed(np.zeros(0), 0).size == 0


def test_failure_reports_the_leading_minor(gpu_ctx):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((300, 310))
    M = A @ A.T / 300 + np.eye(300)
    M[200, 200] = -5.0
    with pytest.raises(DnaGpuError):
        gpu_ctx.cholesky_inverse_packed(pack_lower(M), 300)
    assert gpu_ctx.last_info() == 201       # dpotrf info: first non positive leading minor
    # the context stays usable
    M[200, 200] = 5.0
    inv = unpack_lower(gpu_ctx.cholesky_inverse_packed(pack_lower(M), 300), 300)
    assert np.abs(inv @ M - np.eye(300)).max() < 1e-10


def test_large_inverse_properties(gpu_ctx):
    """size-independent properties at a size the CPU oracle would take minutes for: N * N^-1 = I on random
    probe vectors, symmetry of the result, and idempotence inv(inv(N)) = N"""
    n = 6000
    rng = np.random.default_rng(1)
    m = gpu_ctx.matrix(n)
    m.reset(n)
    ns = n // 3
    blocks = np.tile(np.array([4.0, 1, .5, 1, 5, .25, .5, .25, 6]), ns) * np.repeat(rng.uniform(0.5, 2.0, ns), 9)
    gpu_ctx.add_diag3x3(m, np.arange(ns, dtype=np.uint32), blocks)
    ap0 = m.download_packed()
    m.invert()
    ap1 = m.download_packed()
    x = rng.standard_normal(n)
    y = gpu_ctx.multiply_sym_packed(ap1, gpu_ctx.multiply_sym_packed(ap0, x, n), n)
    assert np.abs(y - x).max() < 1e-11
    m.upload_packed(ap1, n)
    m.invert()
    ap2 = m.download_packed()
    assert np.abs(ap2 - ap0).max() < 1e-11
    m.close()
