"""The 1e-8 m / 1e-8 relative claim pinned WITHOUT passing through the oracle: the device path against the committed extended-precision solution
(tests/golden/exact_3k.npz, made by tools/make_exact_golden.py with tests/exact.py: numpy longdouble -- 64-bit mantissa -- from the assembly of the
normals to a blocked Cholesky of its own) of a 3-block phased GNSS chain of n = 3 600 unknowns: blocks of n = 1 500 (12 tiles: the batched path,
condensed chains with junctions of 300 unknowns, 1e12 constraint weights), once as the library runs it and once with EVERY product on the
128-tile throughput kernel (dnagpu_debug_set_small_tiles(0)).  An fp64 result within a few units in the last place of a 4e6 m coordinate
(9.3e-10 m) of the exact least-squares solution is within twice that of any other such result -- the reference's included."""
import json
import os

import numpy as np
import pytest

from dynadjust_amd import adjust
from tests import fullsize

pytestmark = pytest.mark.gpu

TOL_X = 1e-8
TOL_V = 1e-8


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "exact_3k.npz"))
    return g, json.loads(bytes(g["meta"]).decode())


def compare_with_exact(g, meta, n_blocks, stations_of, estimates_of, variances_of):
    """max |dx| [m] and the largest relative deviations of the variance samples / checksums of a run's blocks from the exact record"""
    x = g["x_hi"].astype(np.longdouble) + g["x_lo"].astype(np.longdouble)
    dx = dv = dvc = dfro = dquad = 0.0
    assert n_blocks == len(meta["block_unknowns"])
    for b in range(n_blocks):
        stn = np.asarray(stations_of(b), dtype=np.int64)
        assert np.array_equal(stn, g[f"stations_{b}"])
        idx = (3 * stn[:, None] + np.arange(3)).ravel()
        est = np.asarray(estimates_of(b), dtype=np.longdouble)
        dx = max(dx, float(np.abs(est - x[idx]).max()))
        var = variances_of(b)
        diag, cols = fullsize.sample_packed(var, idx.size)
        scale = float(np.abs(g[f"vdiag_{b}"]).max())
        dv = max(dv, float(np.abs(diag - g[f"vdiag_{b}"]).max()) / scale)
        dvc = max(dvc, float(np.abs(cols - g[f"vcols_{b}"]).max()) / scale)
        fro, quad = fullsize.packed_checksums(var, idx.size, b)
        dfro = max(dfro, abs(fro - float(g[f"vfro_{b}"][0])) / float(g[f"vfro_{b}"][0]))
        dquad = max(dquad, float(np.abs(quad - g[f"vquad_{b}"]).max() / np.abs(g[f"vquad_{b}"]).max()))
    return {"max_abs_dx_m": dx, "max_rel_dvar_diagonal": dv, "max_rel_dvar_sampled_columns": dvc, "max_rel_dfrobenius": dfro,
            "max_rel_dquadratic_forms": dquad}


@pytest.mark.parametrize("tiles", ["default", "128-tile kernel only"])
@pytest.mark.parametrize("mt", [False, True])
def test_device_against_the_exact_solution_at_size(built, golden_dir, tmp_path, tiles, mt):
    g, meta = _load(golden_dir)
    info = adjust.write_synthetic_network(str(tmp_path), "e", meta["rows"], meta["cols"], 0, meta["blocks"], seed=meta["seed"])
    assert info["stations"] == meta["stations"]
    old = built.dnagpu_debug_set_small_tiles(0) if tiles != "default" else None
    try:
        a = adjust.DnaAdjust()
        a.PrepareAdjustment(adjust.ProjectSettings("e", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=mt))
        st = a.AdjustNetwork()
    finally:
        if old is not None:
            built.dnagpu_debug_set_small_tiles(old)
    assert st == 0 and a.CurrentIteration() == meta["iterations"]
    rec = compare_with_exact(g, meta, a.blockCount(), a.block_stations, a.block_estimates, a.block_variances_packed)
    rec.update(tiles=tiles, chains=4 if mt else 1, unknowns=meta["unknowns"], block_unknowns=meta["block_unknowns"], junction_unknowns=meta["junction_unknowns"])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, f"parity_exact_{'default' if tiles == 'default' else 'tile128'}_{'mt' if mt else 'st'}.json"), "w"), indent=1)
    except OSError:
        pass
    assert rec["max_abs_dx_m"] < TOL_X and rec["max_rel_dvar_diagonal"] < TOL_V and rec["max_rel_dvar_sampled_columns"] < TOL_V, rec
    assert rec["max_rel_dfrobenius"] < TOL_V and rec["max_rel_dquadratic_forms"] < 1e-7, rec
    a.close()
