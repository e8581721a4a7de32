"""Do two inverses on two streams overlap?  Times two n x n inverses run one after the other on chain 0 against the
same two run concurrently on chains 0 and 1 (two host threads; ctypes releases the GIL).  Diagnostic for the
--multi-thread mode (forward || reverse chains): the latency-bound leaves / small GEMMs of one chain can only hide
behind the other chain's big GEMMs if their workgroups find room on a CU."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19968
ctx = DeviceContext(0)
lib = ctx.lib
ns = n // 3
w9 = np.tile(np.array([4.0, 1, .5, 1, 5, .25, .5, .25, 6]), ns)
idx = np.arange(ns, dtype=np.uint32)
mats, keeps = [], []
for c in range(2):
    m = ctx.matrix(n)
    m.reset(3 * ns)
    ctx.add_diag3x3(m, idx, w9)
    k = ctx.matrix(n)
    lib.dnagpu_matrix_copy(ctx.h, 0, k.h, m.h)
    mats.append(m)
    keeps.append(k)
ctx.sync()


def inv(chain, m):
    lib.dnagpu_invert(ctx.h, chain, m.h, 0)


def restore():
    for m, k in zip(mats, keeps):
        lib.dnagpu_matrix_copy(ctx.h, 0, m.h, k.h)
    ctx.sync()


for c in range(2):      # warm-up: plans of both chains
    inv(c, mats[c])
for rep in range(3):
    restore()
    t0 = time.perf_counter()
    inv(0, mats[0])
    inv(0, mats[1])
    seq = time.perf_counter() - t0
    restore()
    th = [threading.Thread(target=inv, args=(c, mats[c])) for c in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    par = time.perf_counter() - t0
    print(f"n={n}: sequential {seq*1e3:.1f} ms, two chains {par*1e3:.1f} ms, ratio {seq/par:.3f}", flush=True)
