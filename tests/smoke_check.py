"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 (device 0), checked against
the CPU oracle.  The oracle is only the checker here; the adjustment itself runs in libdnagpu.so."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from dynadjust_amd import adjust
    from tests import oracle   # test infrastructure, used as the checker only

    d = tempfile.mkdtemp(prefix="dnagpu_smoke_")
    adjust.write_synthetic_network(d, "smoke", 14, 12, 420, 4)
    net = oracle.Network(os.path.join(d, "smoke"), True)
    o = oracle.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(adjust.ProjectSettings("smoke", d, adjust_mode=adjust.PhasedMode))
    st = a.AdjustNetwork()
    assert st == ost == 0, (st, ost)
    worst = 0.0
    for b in range(a.blockCount()):
        worst = max(worst, float(np.abs(a.block_estimates(b) - o.block_estimates(b)).max()))
        vo = o.block_variances(b)
        assert np.abs(a.block_variances_packed(b) - vo).max() / np.abs(vo).max() < 1e-8
    assert worst < 1e-8, worst
    print(f"smoke ok: phased adjustment of {net.n_stations} stations / {a.blockCount()} blocks on the device, "
          f"{a.CurrentIteration()} iterations, max |x_gpu - x_oracle| = {worst:.2e} m")
    a.close()
    o.close()
