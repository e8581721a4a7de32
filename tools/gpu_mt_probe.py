"""diagnostic: the reference's multi-thread schedule on one GPU, many times, while a 3-rank instance of the same network stays alive
(tests/test_gpu_distributed.py::test_ranks_as_threads_sharing_the_gpu[3-False-True]); a run whose results differ from the first is
reported block by block"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from dynadjust_amd import adjust
d = tempfile.mkdtemp()
adjust.write_synthetic_network(d, "n", 30, 12, 0, 6, seed=10)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
schur = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False      # 1: the condensed schedule (the default path) instead of the reference schedule
ref = None
bad = 0
for it in range(N):
    a = None
    if keep:
        a = adjust.DnaAdjust()
        a.PrepareAdjustment(adjust.ProjectSettings("n", d, adjust_mode=adjust.PhasedMode, devices=[0, 0, 0], dist_transport="local", schur_carry=False, multi_thread=True, output_folder=d))
        a.AdjustNetworkDistributed(); a.GenerateStatistics()
    f = adjust.DnaAdjust()
    f.PrepareAdjustment(adjust.ProjectSettings("n", d, adjust_mode=adjust.PhasedMode, schur_carry=schur, multi_thread=True, output_folder=d))
    st = f.AdjustNetwork(); f.GenerateStatistics()
    x = [f.block_estimates(k) for k in range(6)]
    v = [f.block_variances_packed(k) for k in range(6)]
    chi = f.GetChiSquared()
    corr = [f.GetIterationCorrection(i + 1) for i in range(f.CurrentIteration())]
    f.close()
    if a is not None: a.close()
    if it and it % 100 == 0:
        print("runs so far:", it, "deviant:", bad, flush=True)       # (a run under `timeout` still leaves its count)
    if ref is None:
        ref = (x, v, chi); continue
    if chi != ref[2] or any(not np.array_equal(x[k], ref[0][k]) for k in range(6)):
        bad += 1
        print("run", it, "chi2 %.10f vs %.10f" % (chi, ref[2]), "corr", corr, "per block max|dx|", ["%.2e" % np.abs(x[k] - ref[0][k]).max() for k in range(6)],
              "max|dvar|/max", ["%.2e" % (np.abs(v[k] - ref[1][k]).max() / np.abs(ref[1][k]).max()) for k in range(6)], flush=True)
print("deviant runs:", bad, "of", N - 1)
