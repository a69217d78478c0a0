"""How many columns iterate to itermax per step in a long run (census of the pass counts, a launch per step), and what a
launch of many steps then takes per step.  python tools/r05_census.py nz ncol settle nsingle nmulti"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 260
nsingle = int(sys.argv[4]) if len(sys.argv) > 4 else 12
nmulti = int(sys.argv[5]) if len(sys.argv) > 5 else 100
idx = np.arange(0, 100000, 100000 // ncol)[:ncol]
kc, k3 = cm.make_hip_case(len(idx), nz, index=idx, ntotal=100000)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(len(idx), "bench", index=idx)); ctx.set_forcing(k3.sflux)
ctx.step(1, settle); ctx.synchronize()
nt = settle + 1
print(f"nz={nz} ncol={len(idx)} after {settle} steps (MCKPP_SOLO={os.environ.get('MCKPP_SOLO', 'default')}):", flush=True)
for _ in range(nsingle):
    ctx.step(nt, 1); ctx.synchronize()
    ms, _n = ctx.last_kernel_ms()
    st, nf, npass = ctx.status()
    big = np.flatnonzero(npass > 50)
    print(f"  step {nt}: {ms:.3f} ms, mean passes {npass.mean():.2f}, max {npass.max()}, columns > 12 passes: {(npass > 12).sum()}, > 50: {len(big)} {list(idx[big][:6])}", flush=True)
    nt += 1
t0 = time.perf_counter()
ctx.step(nt, nmulti); ctx.synchronize()
dt = time.perf_counter() - t0
ms, _n = ctx.last_kernel_ms()
print(f"  {nmulti} steps in one launch: {dt / nmulti * 1e3:.3f} ms per step (kernel {ms / nmulti:.3f})", flush=True)
ctx.close()
