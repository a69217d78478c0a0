import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm
import mckpp_f90_amd as mk
ncol, nz = 100000, 100
seq = [int(x) for x in sys.argv[1].split(",")]
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.set_diagnostics(1); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
nt = 1
out = []
for n in seq:
    t0 = time.perf_counter(); ctx.step(nt, n); ctx.synchronize(); dt = time.perf_counter() - t0
    st, nf, npass = ctx.status()
    out.append(f"{nt}..{nt+n-1}: {dt/n*1e3:.2f} ms (last: max {int(npass.max())}, >12: {int((npass>12).sum())}, >50: {int((npass>50).sum())})")
    nt += n
print(" | ".join(out))
ctx.close()
