"""Kernel time and pass counts of every step of a long run of the headline workload (one launch per step): are
the slow steps of a sustained run slow because of the clock, or because a column runs to itermax?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402

ncol, nz = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 60
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 460
kc, k3 = cm.make_hip_case(ncol, nz)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
rows = []
for nt in range(1, nsteps + 1):
    ctx.step(nt, 1); ctx.synchronize()
    ms, nl = ctx.last_kernel_ms()
    st, nf, npass = ctx.status()
    rows.append((nt, ms, float(npass.mean()), int(npass.max()), int((npass > 12).sum())))
a = np.array([r[1] for r in rows]); mx = np.array([r[3] for r in rows])
print(f"nz={nz}: steps 10..{nsteps}: kernel ms median {np.median(a[9:]):.3f}, mean {a[9:].mean():.3f}")
slow = mx[9:] > 50
print(f"  steps with a column of more than 50 passes: {int(slow.sum())} of {len(slow)}: mean kernel {a[9:][slow].mean() if slow.any() else 0:.3f} ms; the others {a[9:][~slow].mean():.3f} ms")
for lo in range(10, nsteps, 50):
    seg = slice(lo - 1, min(lo + 49, nsteps))
    print(f"  steps {lo}-{min(lo + 49, nsteps)}: mean {a[seg].mean():.3f} ms, without the long-column steps {a[seg][mx[seg] <= 50].mean():.3f} ms, long-column steps {int((mx[seg] > 50).sum())}, max passes {int(mx[seg].max())}")
print("  first steps (nt, ms, mean passes, max passes, columns > 12 passes):", rows[:8])
ctx.close()
