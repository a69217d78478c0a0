"""Rates of the three builds of the column kernel on 1e5 columns x NZ levels: default physics, optional physics
(k_column_ps<EXT> with L_DAMP_CURR + L_NO_FREEZE on - no extra input fields needed) and optional physics with double
diffusion (LDD: the 15-row variant, no L1 under the V sweep).  Run on the GPU box: python tools/r03_ext_rate.py [nz]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (before the library: one HIP runtime per process)

import common as cm   # noqa: E402

import mckpp_f90_amd as mk   # noqa: E402

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ncol = 100000
for tag, sw in (("default", {}), ("optional physics", dict(L_DAMP_CURR=1, L_NO_FREEZE=1)), ("double diffusion", dict(LDD=1))):
    kc, k3 = cm.make_hip_case(ncol, nz)
    for k, v in sw.items():
        setattr(kc, k, v)
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench"))
    ctx.set_forcing(k3.sflux)
    ctx.step(1, 6)            # spin-up from the analytic profile
    ctx.synchronize()
    ctx.step(7, 20)
    ctx.synchronize()
    ms, n = ctx.last_kernel_ms()
    b, mx, t, lds = ctx.kernel_residency()
    print(f"{tag:17s} {ctx.kernel_name:18s} nz={nz}: {ncol * n / (ms * 1e-3):.3e} column-steps/s ({ms / n:.3f} ms per step), "
          f"{b} workgroup(s) of {t} threads and {lds} B of LDS per CU")
    ctx.close()
    kc._hip_ctx = None
