import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from dynadjust_amd import adjust
def stats(a):
    return np.array([a.GetChiSquared(), a.GetSigmaZero(), a.GetGlobalPelzerRel(), float(a.GetPotentialOutlierCount()), float(a.GetDegreesOfFreedom()), float(a.GetTestResult())])
def run(folder, name, **kw):
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, **kw)
    a = adjust.DnaAdjust(); a.PrepareAdjustment(p); return a
d = tempfile.mkdtemp()
adjust.write_synthetic_network(d, "n", 30, 12, 0, 6, seed=10)
for rep in range(3):
  for ranks, schur, mt in [(2, True, False), (2, False, False), (3, True, True), (3, False, True), (8, True, False)]:
    a = run(d, "n", devices=[0] * ranks, dist_transport="local", schur_carry=schur, multi_thread=mt, output_folder=d)
    a.AdjustNetworkDistributed(); a.GenerateStatistics(); sa = stats(a)
    xa = [a.block_estimates(k) for k in range(6)]
    a.close()
    f = run(d, "n", schur_carry=schur, multi_thread=mt, output_folder=d)
    f.AdjustNetwork(); f.GenerateStatistics(); sf = stats(f)
    xf = [f.block_estimates(k) for k in range(6)]
    f.close()
    print(rep, ranks, schur, mt, "chi2 rel diff %.3e" % (abs(sa[0]-sf[0])/sf[0]), "max |dx| %.3e" % max(np.abs(xa[k]-xf[k]).max() for k in range(6)), "chi2 %.10f %.10f" % (sa[0], sf[0]), flush=True)
