"""ms per step of mckpp_hip_step(nt, n) as a function of n (one launch for n steps), after a spin-up."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402

ncol, nz = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 60
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
ctx.step(1, 12); ctx.synchronize()
nt = 13
for n in (1, 1, 2, 5, 10, 20, 40, 100, 40, 10, 1):
    t0 = time.perf_counter(); ctx.step(nt, n); ctx.synchronize(); dt = time.perf_counter() - t0
    ms, nl = ctx.last_kernel_ms()
    st, nf, npass = ctx.status()
    nt += n
    print(f"nz={nz} n={n:4d}: {dt / n * 1e3:.3f} ms per step (kernel {ms / nl:.3f}), {ctx.last_launch_count()} launch(es), last step max passes {int(npass.max())}, steps so far {nt - 1}")
ctx.close()
