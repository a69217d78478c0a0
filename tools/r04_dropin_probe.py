"""Where the drop-in step's time goes: set_forcing / step / download(scalar group) timed apart (a synchronisation
after each), in a fresh context right after the kernel-only legs and again after 400 more steps."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402


def loop(ctx, k3, nt, n, mask):
    rows = []
    for _ in range(n):
        t0 = time.perf_counter(); ctx.set_forcing(k3.sflux); ctx.synchronize()
        t1 = time.perf_counter(); ctx.step(nt, 1); ctx.synchronize()
        t2 = time.perf_counter(); ctx.download(k3, mask)
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2))
        nt += 1
    a = np.array(rows) * 1e3
    return nt, a


def fused(ctx, k3, nt, n, mask):
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.set_forcing(k3.sflux); ctx.step(nt, 1); ctx.download(k3, mask); nt += 1
    return nt, (time.perf_counter() - t0) / n * 1e3


ncol, nz = 100000, 60
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
ctx.step(1, 46); ctx.synchronize()
nt = 47
mask = mk.api.F_SCALARS
ctx.set_forcing(k3.sflux); ctx.step(nt, 1); ctx.download(k3, mask); nt += 1     # pins the arrays
for tag in ("after 46 kernel-only steps", "after 400 more kernel-only steps", "after 20 idle ms"):
    nt, a = loop(ctx, k3, nt, 10, mask)
    print(f"{tag}: set_forcing {a[:,0].mean():.2f} ms, step {a[:,1].mean():.2f} ms, download {a[:,2].mean():.2f} ms  (first three downloads: "
          + ", ".join(f"{x:.2f}" for x in a[:3, 2]) + ")")
    nt, f = fused(ctx, k3, nt, 20, mask)
    print(f"    the three calls back to back, one wait per step: {f:.2f} ms per step")
    if tag.startswith("after 46"):
        ctx.step(nt, 400); ctx.synchronize(); nt += 400
    else:
        time.sleep(0.02)
ctx.close()
