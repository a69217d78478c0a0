import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm
import mckpp_f90_amd as mk
from oracle import orc
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ncol = 45
oc, ob = cm.make_oracle(ncol, nz, init=False, exp_mode=1)
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.mckpp_initialize_ocean_model(k3, kc)
orc.init_ocean(oc, ob, 0)
bad = np.arange(0, ncol, 4)
k3.U[bad, 0:4, 0] = 50.0
ob["U"][bad, 1:5] = 50.0
ctx.upload(k3)
sf = cm.synth.forcing(ncol, "bench")
ob["sflux"] = sf
cm.set_forcing_3d(k3, sf)
for nt in (1, 2):
    mk.mckpp_physics_driver(k3, kc, nt)
    orc.physics_driver(oc, ob, nt)
    st, nf, npass = ctx.status()
    print("step", nt, ctx.kernel_name, "status eq", np.array_equal(st, ob["status"]), "npass eq", np.array_equal(npass, ob["npasses"]))
    dU = np.abs(k3.U[:, :, 0] - ob["U"][:, 1:nz + 2])
    cols = np.nonzero(dU.max(axis=1) > 0)[0]
    print(" differing columns", cols, "status", st[cols], "npass", npass[cols], "in bad:", np.isin(cols, bad))
    for c in cols:
        print("  col", c, "levels", np.nonzero(dU[c] > 0)[0], "hmix", k3.hmix[c], ob["hmix"][c], "kmix", k3.kmix[c], ob["kmix"][c], "reset", k3.reset_flag[c], ob["reset_flag"][c])
