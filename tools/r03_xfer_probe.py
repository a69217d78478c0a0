"""Where a download's time goes (1e5 x 60): per field group, repeated; run under rocprofv3 --kernel-trace for the
layout kernels' durations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (before the library)
import common as cm
import mckpp_f90_amd as mk

ncol, nz = 100000, 60
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
ctx.step(1, 3); ctx.synchronize()
for name, mask, nf in (("profiles", mk.api.F_PROFILES, 4), ("saved", mk.api.F_SAVED, 8), ("scalars", mk.api.F_SCALARS, 0),
                       ("restart", mk.api.F_RESTART, 12), ("diag", mk.api.F_DIAG, 18)):
    ctx.download(k3, mask)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.download(k3, mask); ts.append(time.perf_counter() - t0)
    t = min(ts)
    mb = nf * ncol * (nz + 1) * 8 / 1e6
    print(f"{name:9s} {t*1e3:7.2f} ms" + (f"  {mb:6.0f} MB -> {mb/1e3/t:5.1f} GB/s, {t*1e3/nf:5.2f} ms per field" if nf else ""))
t0 = time.perf_counter()
for _ in range(20):
    sc = k3.as_c()
print(f"as_c() {(time.perf_counter()-t0)/20*1e3:.3f} ms")
# the drop-in loop, its parts timed with a wait after each
nt = 4
for name, mask in (("restart", mk.api.F_RESTART),):
    acc = {"forcing": 0.0, "step": 0.0, "download": 0.0}
    for it in range(6):
        nt += 1
        t0 = time.perf_counter(); ctx.set_forcing(k3.sflux); ctx.synchronize(); t1 = time.perf_counter()
        ctx.step(nt, 1); ctx.synchronize(); t2 = time.perf_counter()
        ctx.download(k3, mask); t3 = time.perf_counter()
        acc["forcing"] += t1 - t0; acc["step"] += t2 - t1; acc["download"] += t3 - t2
    print(name, {k: round(v / 6 * 1e3, 3) for k, v in acc.items()})
    t0 = time.perf_counter()
    for it in range(6):
        nt += 1
        ctx.set_forcing(k3.sflux); ctx.step(nt, 1); ctx.download(k3, mask)
    print(name, "no waits in between: %.3f ms per step" % ((time.perf_counter() - t0) / 6 * 1e3))
# one step per call with a host wait after each, against many steps per call (same work)
ctx.step(nt + 1, 20); ctx.synchronize(); nt += 20
t0 = time.perf_counter(); ctx.step(nt + 1, 20); ctx.synchronize(); t1 = time.perf_counter(); nt += 20
print("20 steps in one call: %.3f ms per step" % ((t1 - t0) / 20 * 1e3))
t0 = time.perf_counter()
for it in range(20):
    nt += 1; ctx.step(nt, 1); ctx.synchronize()
print("20 calls of one step, a wait after each: %.3f ms per step; kernel %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3, ctx.last_kernel_ms()[0]))
t0 = time.perf_counter()
for it in range(20):
    nt += 1; ctx.set_forcing(k3.sflux); ctx.step(nt, 1); ctx.synchronize()
print("  + set_forcing: %.3f ms per step; kernel %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3, ctx.last_kernel_ms()[0]))
t0 = time.perf_counter()
for it in range(20):
    nt += 1; ctx.set_forcing(k3.sflux); ctx.step(nt, 1); ctx.download(k3, mk.api.F_SCALARS)
print("  + download of the scalar group: %.3f ms per step; kernel %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3, ctx.last_kernel_ms()[0]))
