#!/usr/bin/env python3
"""tools/r04_tolerance.py - what tolerance holds between the device path and the reference's arithmetic.

CPU only.  The HIP kernels are bit-identical to the oracle run with exp_mode=1 (asserted by the -m gpu tests on
every shape), so oracle(exp_mode=1) against the other oracle variants IS the table of the device path:

  hip           exp_mode=1                      the bits mckpp_hip_step produces (reference-order solver)
  faithful      exp_mode=0                      libm exp = the reference's EXP as amdflang builds it
  faithful_pow  exp_mode=0, half_pow_mode=1     the reference as a compiler that lowers x**(1./2.) to pow() would
                                                build it (28 entries of wst differ, lookup_mod.F90:60-62)
  hip_2e        exp_mode=1, solver_mode=1       the library's opt-in two-ended tridiagonal elimination

The variants are stepped side by side over the BASELINE config shapes; at every step the columns whose kmix differs
are counted, at the checkpoints the SURVEY 8(d) parity-gate figures are taken (tests/common.py: tolerance_metrics).
Writes profiles/r04/parity_tolerance.json.   python tools/r04_tolerance.py [--quick] [--only cfg2] [--threads N]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import common as cm  # noqa: E402
from oracle import orc  # noqa: E402

VARIANTS = {
    "hip": dict(exp_mode=1),
    "faithful": dict(exp_mode=0),
    "faithful_pow": dict(exp_mode=0, half_pow_mode=1),
    "hip_2e": dict(exp_mode=1, solver_mode=1),
}
PAIRS = [("hip", "faithful"), ("faithful_pow", "faithful"), ("hip", "faithful_pow"), ("hip_2e", "hip"),
         ("hip_2e", "faithful")]
NAMES = ("taux", "tauy", "swf", "lwf", "lhf", "shf", "rain", "snow")


def run_case(tag, ncol, ntotal, nz, grid, dto, nsteps, checkpoints, diurnal, threads, land_frac=0.0, winds=False):
    idx = np.arange(ncol) * (ntotal // ncol) if ncol < ntotal else np.arange(ncol)
    t0 = time.time()
    runs = {}
    for v, kw in VARIANTS.items():
        oc, ob = cm.make_oracle(ncol, nz, mix="bench", grid=grid, dto=dto, index=idx, ntotal=ntotal, nthreads=threads, **kw)
        runs[v] = (oc, ob)
    active = np.ones(ncol, bool)
    if land_frac > 0:   # the land points of the configs[4] shape take no part (run_physics = .F.)
        active = ~((np.arange(ncol) * 7) % 20 < int(round(20 * land_frac)))
        for oc, ob in runs.values():
            ob["l_ocean"] = active.astype(np.int32)
    same = {p: active.copy() for p in PAIRS}          # columns whose kmix and pass count agreed at every step so far
    flip_steps = {p: 0 for p in PAIRS}                # column-steps with differing kmix
    first_flip = {p: None for p in PAIRS}
    pass_diff = {p: 0 for p in PAIRS}
    first_pass_diff = {p: None for p in PAIRS}
    # winds: the bench mix has two wind stresses only, i.e. two values of ustar, i.e. four columns of the wmt/wst
    # table; this case spreads taux over 0..1.6 N/m2 (ustar over the table's whole 0..0.04 m/s) column by column
    taux_w = 1.6 * ((np.asarray(idx) * 0.6180339887498949) % 1.0) ** 2 if winds else None
    if winds and not diurnal:
        for oc, ob in runs.values():
            ob["sflux"][:, 0] = taux_w
    table = []
    for nt in range(1, nsteps + 1):
        if diurnal:
            ser = cm.synth.flux_series(ncol, nt, 1, dto, "bench", idx)[0]
            if winds:
                ser[0] = taux_w
        for oc, ob in runs.values():
            if diurnal:
                orc.fluxes(oc, ob, nt, **dict(zip(NAMES, ser)))
            orc.physics_driver(oc, ob, nt, nthreads=threads)
        for p in PAIRS:
            a, b = runs[p[0]][1], runs[p[1]][1]
            d = (a["kmix"] != b["kmix"]) & active
            n = int(d.sum())
            flip_steps[p] += n
            dp = (a["npasses"] != b["npasses"]) & active
            pass_diff[p] += int(dp.sum())
            if n and first_flip[p] is None:
                first_flip[p] = nt
            if dp.any() and first_pass_diff[p] is None:
                first_pass_diff[p] = nt
            same[p] &= ~d & ~dp
        if nt in checkpoints:
            for p in PAIRS:
                m = cm.tolerance_metrics(cm.oracle_state(runs[p[0]][1], nz), cm.oracle_state(runs[p[1]][1], nz), same[p], active)
                m.update({"pair": f"{p[0]} vs {p[1]}", "step": nt,
                          "column_steps_with_other_kmix_per_1e5": flip_steps[p] / (active.sum() * nt) * 1e5,
                          "column_steps_with_other_pass_count_per_1e5": pass_diff[p] / (active.sum() * nt) * 1e5,
                          "first_step_with_a_kmix_difference": first_flip[p],
                          "first_step_with_a_pass_count_difference": first_pass_diff[p]})
                table.append(m)
            print(f"[{tag}] step {nt}: " + "; ".join(
                f"{r['pair']}: off-path {r['off_path_columns']}, hmix {r['same_path']['hmix']['max']:.1e}, "
                f"T {r['same_path']['T']['max']:.1e}, U {r['same_path']['U']['max']:.1e}"
                for r in table[-len(PAIRS):]) + f"  ({time.time() - t0:.0f} s)", flush=True)
    npass = {v: float(ob["npasses"][active].mean()) for v, (oc, ob) in runs.items()}
    return {"case": tag, "columns": int(active.sum()), "of": ntotal, "levels": nz, "grid": grid, "dto": dto,
            "forcing": ("bench mix, diurnal short-wave cycle (mckpp_fluxes every step)" if diurnal else "bench mix, constant")
                       + (", taux spread over 0..1.6 N/m2 column by column" if winds else ""),
            "steps": nsteps, "mean_passes_last_step": npass, "seconds": time.time() - t0, "rows": table}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="a tenth of the columns, fewer steps (smoke run)")
    ap.add_argument("--only", default="")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "parity_tolerance.json"))
    a = ap.parse_args()
    q = a.quick
    cases = [
        ("cfg0", dict(ncol=64, ntotal=64, nz=40, grid="uniform", dto=3600.0, nsteps=1, checkpoints=[1], diurnal=False)),
        ("cfg2", dict(ncol=10000 if q else 100000, ntotal=100000, nz=60, grid="uniform", dto=3600.0,
                      nsteps=24 if q else 72, checkpoints=[1, 3, 24, 72], diurnal=True)),
        ("cfg2_winds", dict(ncol=2000 if q else 20000, ntotal=100000, nz=60, grid="uniform", dto=3600.0,
                            nsteps=24 if q else 72, checkpoints=[1, 3, 24, 72], diurnal=True, winds=True)),
        ("cfg3", dict(ncol=250, ntotal=100000, nz=100, grid="uniform", dto=3600.0, nsteps=48 if q else 1000,
                      checkpoints=[24, 48, 120, 240, 360, 480, 600, 720, 860, 1000], diurnal=True)),
        ("cfg3_more_columns", dict(ncol=250 if q else 4000, ntotal=100000, nz=100, grid="uniform", dto=3600.0,
                                   nsteps=48 if q else 240, checkpoints=[24, 48, 120, 240], diurnal=True)),
        ("cfg4", dict(ncol=3000 if q else 30000, ntotal=144507, nz=69, grid="stretched", dto=1200.0,
                      nsteps=24 if q else 72, checkpoints=[1, 3, 24, 72], diurnal=True, land_frac=0.35)),
    ]
    out = {"what": __doc__.split("\n\n")[1], "error_definition":
           "hmix: |a-b|/|b|; T,S,U,V: max over the levels of |a-b| / max over the levels of |b| (S is the anomaly "
           "from Sref); same_path = columns whose kmix and pass count agreed at every step so far, off_path = the others",
           "variants": VARIANTS, "cases": []}
    for tag, kw in cases:
        if a.only and tag not in a.only.split(","):
            continue
        out["cases"].append(run_case(tag, threads=a.threads, **kw))
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
