"""bench.py's side_shape for 100 levels, call by call: where does the time of its 10 timed steps go?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402

ncol, nz = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.set_diagnostics(1); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
nt = 1
for n in (5, 10, 10, 10):
    t0 = time.perf_counter(); ctx.step(nt, n); ctx.synchronize(); dt = time.perf_counter() - t0
    ms, nl = ctx.last_kernel_ms()
    st, nf, npass = ctx.status()
    nt += n
    print(f"nz={nz} steps {nt - n}..{nt - 1}: {dt / n * 1e3:.3f} ms per step (kernel {ms / nl:.3f}), {ctx.last_launch_count()} launch(es), last step: mean passes {npass.mean():.3f} max {int(npass.max())}, columns > 12 passes {int((npass > 12).sum())}")
ctx.close()
