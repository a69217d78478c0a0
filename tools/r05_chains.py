"""The chains of a long run: the steps of a launch of many steps one by one, the pass count of every column in every step,
and from them what the launch's longest chain (one column's passes, step after step) is against the device's throughput.
python tools/r05_chains.py nz ncol settle nsteps"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 60
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
idx = np.arange(0, 100000, 100000 // ncol)[:ncol]
kc, k3 = cm.make_hip_case(len(idx), nz, index=idx, ntotal=100000)
ctx = mk.MckppHip(kc)
ctx.upload(k3); ctx.init_ocean(0)
cm.set_forcing_3d(k3, cm.synth.forcing(len(idx), "bench", index=idx)); ctx.set_forcing(k3.sflux)
ctx.step(1, settle); ctx.synchronize()
nt = settle + 1
tot = np.zeros(len(idx), dtype=np.int64)
nit = np.zeros(len(idx), dtype=np.int64)
per_step = []
hist = []
for _ in range(nsteps):
    ctx.step(nt, 1); ctx.synchronize()
    st, nf, npass = ctx.status()
    tot += npass
    nit += npass > 50
    per_step.append(int((npass > 50).sum()))
    hist.append(npass.copy())
    nt += 1
order = np.argsort(-tot)
print(f"nz={nz} ncol={len(idx)}: {nsteps} steps after {settle}; columns over 50 passes per step: mean {np.mean(per_step):.1f} max {max(per_step)}")
print(f"  passes of a column over the {nsteps} steps: mean {tot.mean():.0f}, longest chains {list(tot[order][:8])} (steps at itermax {list(nit[order][:8])})")
hist = np.asarray(hist)
for r in range(3):
    print(f"  column {idx[order[r]]}: steps (of the {nsteps}) at itermax: {list(np.flatnonzero(hist[:, order[r]] > 50))}")
for thr in (1, 5, 10, 20, 40):
    print(f"  columns with >= {thr} steps at itermax: {(nit >= thr).sum()}")
for us in (43.5, 26.0, 19.0):
    print(f"  longest chain at {us} us per pass: {tot.max() * us * 1e-3:.0f} ms = {tot.max() * us * 1e-3 / nsteps:.3f} ms per step")
print(f"  all passes / (slots x ...): total passes {tot.sum()}")
ctx.close()


def alone(cols, label):
    """the same steps, one launch, for these columns only"""
    import time
    sub = np.asarray(sorted(cols))
    kc2, k32 = cm.make_hip_case(len(sub), nz, index=sub, ntotal=100000)
    c2 = mk.MckppHip(kc2)
    c2.upload(k32); c2.init_ocean(0)
    cm.set_forcing_3d(k32, cm.synth.forcing(len(sub), "bench", index=sub)); c2.set_forcing(k32.sflux)
    c2.step(1, settle); c2.synchronize()
    t0 = time.perf_counter()
    c2.step(settle + 1, nsteps); c2.synchronize()
    dt = time.perf_counter() - t0
    st, nf, npass = c2.status()
    print(f"  {label}: {len(sub)} columns alone, {nsteps} steps in one launch: {dt * 1e3:.1f} ms = {dt / nsteps * 1e3:.3f} ms per step", flush=True)
    c2.close()


if os.environ.get("ALONE", "1") == "4":
    reg = nit == 0
    for n in (7, 50, 156):
        sel = reg.copy(); sel[order[:n]] = True
        alone(idx[sel], f"the columns never at itermax + the {n} longest chains (MCKPP_GATHER={os.environ.get('MCKPP_GATHER')})")
elif os.environ.get("ALONE", "1") == "3":
    for k in (1, 2, 4, 9, 19):
        alone(idx[order[:k]], f"the {k} longest in one workgroup (MCKPP_PS={os.environ.get('MCKPP_PS')})")
        print(f"    their passes: {list(tot[order[:k]])}")
elif os.environ.get("ALONE", "1") == "2":
    alone(idx, "every column")
    alone(idx[nit < 20], "all but those with >= 20 steps at itermax")
    alone(idx[nit < 10], "all but those with >= 10")
    alone(idx[nit < 5], "all but those with >= 5")
    alone(idx[nit < 1], "all but those with >= 1")
elif os.environ.get("ALONE", "1") != "0":
    alone(idx[order[:1]], "the longest chain")
    alone(idx[order[:8]], "the 8 longest")
    alone(idx[order[:64]], "the 64 longest")
    alone(idx[nit >= 1], "every column that is at itermax in some step")
    alone(idx[nit == 0], "every column that never is")
