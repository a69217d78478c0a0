"""Rate of the optional-physics kernel (SST relaxation + salinity relaxation on): python tools/r05_options_rate.py nz [ncol]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import common as cm
import mckpp_f90_amd as mk
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
kc, k3 = cm.make_hip_case(ncol, nz)
kc.L_RELAX_SST = 1; kc.L_RELAX_SAL = 1
k3.relax_sst[:] = 1.0 / (5 * 86400.0); k3.SST0[:] = k3.X[:, 0, 0] + 0.5
k3.relax_sal[:] = 1.0 / (30 * 86400.0); k3.sal_clim[:, :] = k3.X[:, :, 1]
ctx = mk.MckppHip(kc); ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
ctx.step(1, 103); ctx.synchronize()
for r in range(3):
    ctx.step(104 + 20 * r, 20); ctx.synchronize()
    ms, n = ctx.last_kernel_ms()
    st, nf, npass = ctx.status()
    print(f"nz={nz} passes mean {npass.mean():.2f} max {npass.max()} optional physics (MCKPP_PS_FIXED_L={os.environ.get('MCKPP_PS_FIXED_L', '1')}): {ms / 20:.3f} ms per step")
ctx.close()
