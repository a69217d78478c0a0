"""What does one pass of a column that iterates to itermax cost when nothing else shares its workgroup?

From the analytic start profile the second model step takes 14 % of the columns to itermax (200 passes).  One such
column alone in a context (ncol = 1: the launcher gives it a one-slot workgroup of two waves), and alone in the
geometry a full launch uses (MCKPP_PS forced), at 60 and 100 levels, both solver modes: kernel time of step 2 / passes.
Usage: python tools/r05_lone_probe.py [nz ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
import mckpp_f90_amd as mk  # noqa: E402

NTOTAL = 100000


def run(idx, nz, solver_mode=0, steps=2):
    idx = np.asarray(idx)
    kc, k3 = cm.make_hip_case(len(idx), nz, index=idx, ntotal=NTOTAL)
    ctx = mk.MckppHip(kc)
    ctx.set_solver_mode(solver_mode)
    ctx.upload(k3); ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(len(idx), "bench", index=idx)); ctx.set_forcing(k3.sflux)
    out = []
    for nt in range(1, steps + 1):
        ctx.step(nt, 1); ctx.synchronize()
        ms, _ = ctx.last_kernel_ms()
        st, nf, npass = ctx.status()
        out.append((ms, npass.copy()))
    ctx.close()
    return out


for nz in [int(x) for x in sys.argv[1:]] or [60, 100]:
    sample = np.arange(0, NTOTAL, 50)
    r = run(sample, nz)
    npass2 = r[1][1]
    long_cols = sample[npass2 >= 200]
    print(f"nz={nz}: {len(long_cols)} of {len(sample)} sampled columns at itermax in step 2; first: {long_cols[:4]}", flush=True)
    c = int(long_cols[0])
    short = int(sample[npass2 <= 8][0]) if (npass2 <= 8).any() else None
    for sm in (0, 1):
        for geom in (None, "15x8x2" if nz <= 69 else "19x16x1", "1x16x1", "1x8x1", "1x4x1"):
            if geom: os.environ["MCKPP_PS"] = geom
            else: os.environ.pop("MCKPP_PS", None)
            reps = [run([c], nz, sm) for _ in range(int(os.environ.get("LONE_REPS", "1")))]
            a = min(reps, key=lambda r: r[1][0])
            ms, npass = a[1]
            print(f"  nz={nz} solver={sm} geometry={geom or 'launcher (ncol=1)'}: step 2 {ms:.3f} ms, {int(npass[0])} passes -> "
                  f"{ms / npass[0] * 1e3:.2f} us per lone pass (best of {len(reps)}: " + " ".join(f"{r[1][0] / r[1][1][0] * 1e3:.2f}" for r in reps)
                  + f"); step 1 {a[0][0]:.3f} ms / {int(a[0][1][0])} passes", flush=True)
    os.environ.pop("MCKPP_PS", None)
