"""End-to-end parity: product (HIP) vs CPU oracle on synthetic networks (run through gpurun)."""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynadjust_amd import adjust
from dynadjust_amd.device import unpack_lower
from tests import oracle

def run_case(rows, cols, nbl, blocks, phased, scale=False, mt=False):
    d = tempfile.mkdtemp()
    info = adjust.write_synthetic_network(d, "net", rows, cols, nbl, blocks)
    base = d + "/net"
    net = oracle.Network(base, phased)
    t0 = time.time()
    o = oracle.Adjustment(net, phased, scale_normals_to_unity=scale); o.prepare(); ost = o.run()
    t_or = time.time() - t0
    p = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode if phased else adjust.SimultaneousMode,
                               scale_normals_to_unity=scale, multi_thread=mt)
    a = adjust.DnaAdjust()
    t0 = time.time()
    a.PrepareAdjustment(p)
    t_prep = time.time() - t0
    st = a.AdjustNetwork()
    assert st == ost, (st, ost)
    assert a.CurrentIteration() == o.iterations(), (a.CurrentIteration(), o.iterations())
    dx = 0; dv = 0
    for b in range(a.blockCount()):
        assert np.array_equal(a.block_stations(b), o.block_stations(b))
        xe = a.block_estimates(b); xo = o.block_estimates(b)
        dx = max(dx, np.abs(xe - xo).max())
        ve = a.block_variances_packed(b); vo = o.block_variances(b)
        dv = max(dv, np.abs(ve - vo).max() / np.abs(vo).max())
    corr = [(a.GetIterationCorrection(i + 1), o.max_correction(i + 1)) for i in range(o.iterations())]
    print(f"{'phased' if phased else 'simult'} {rows}x{cols} bl={info['baselines']} blocks={blocks} scale={scale} mt={mt}: status {st} iters {a.CurrentIteration()} "
          f"max|dx|={dx:.3e} m  max rel dV={dv:.3e}  corr={corr}  adjust {a.adjustTime():.1f} ms prep {t_prep*1e3:.0f} ms oracle {t_or*1e3:.0f} ms", flush=True)
    a.close(); o.close()
    return dx, dv

ok = True
for case in [(6, 5, 0, 1, False), (12, 10, 300, 1, False), (12, 10, 300, 4, True), (12, 10, 300, 2, True), (12, 10, 300, 3, True, True),
             (30, 30, 2400, 5, True), (40, 25, 0, 1, False, True), (60, 60, 0, 6, True)]:
    dx, dv = run_case(*case)
    ok &= dx < 1e-8 and dv < 1e-8
print("PARITY OK" if ok else "PARITY FAILED")
