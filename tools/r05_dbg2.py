"""debug: forced 3-step launch of 701 columns x 40 levels, state saved for comparison between MCKPP_GATHER settings"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import common as cm
import mckpp_f90_amd as mk
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_multi_gpu import _series
ncol, nz, ndtocn, nsteps = 701, 40, 2, 5
series = _series(ncol, (nsteps + 3 + ndtocn - 1) // ndtocn + 1, 11)
kc, k3 = cm.make_hip_case(ncol, nz, grid="uniform", land_every=5)
h = mk.MckppHip(kc)
h.upload(k3); h.init_ocean(0); h.set_flux_series(0, series)
for nt in range(1, nsteps + 1):
    h.run_forced(nt, 1, ndtocn)
multi = int(os.environ.get("DBG_MULTI", "1"))
if multi:
    h.run_forced(nsteps + 1, 3, ndtocn)
    h.synchronize()
else:
    for nt in range(nsteps + 1, nsteps + 4):
        h.run_forced(nt, 1, ndtocn)
h.download(k3, mk.api.F_RESTART)
st, nf, npass = h.status()
np.savez(sys.argv[1], U=k3.U, X=k3.X, old=k3.old, new=k3.new_, hmix=k3.hmix, npass=npass, run=k3.run_physics)
