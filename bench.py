#!/usr/bin/env python3
"""bench.py - column-steps/s of the MC-KPP column-physics step on MI355X.

Headline workload (BASELINE.json configs[2]): 1e5 synthetic columns x 60 levels per GPU
(spun up for SPINUP model steps, see below), full ocnstep (KPP mixing stack with swfrac + equation of
state, tridiagonal solves), fp64, state resident in HBM.  A "step" is one mckpp_physics_driver
call over the rank's columns.  Columns shard across GPUs with no data-path
collective (weak scaling: 1e5 columns per GPU); a torch.distributed (RCCL)
gather of hmix to rank 0 runs after the timed region only, as the diagnostics
gather the path has.

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`, rendezvous on 127.0.0.1) before this
process has touched torch or the GPU, and relays rank 0's line and exit code; under a launcher (RANK /
WORLD_SIZE set) it is one of the ranks.  `--total-ncol T` divides T columns over the ranks instead
(strong scaling, e.g. `--total-ncol 100000 --nz 100` = BASELINE configs[3]).  The N > 1 line also carries the
per-rank step times, the RCCL world size seen by a device all-reduce, the time of the diagnostics gather (hmix,
T) and a `single_process` block: the same shards behind ONE handle of the C-ABI (mckpp_hip_multi_*), all N GPUs
driven from rank 0's process while the other ranks wait.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`,
plus (N=1 only, never part of `value`):
  burst                the K steps as a cold device runs them (before the settle leg; `value` is measured after
                       SETTLE steps of the same work, i.e. at the clock a model run sees)
  sustained            400 more steps of the headline workload in one call
  two_ended_solver     the headline workload with mckpp_hip_set_solver_mode(1) (opt-in: every tridiagonal system
                       eliminated from both ends at once; `value` itself is the reference-order solver)
  config1_pass         BASELINE configs[1]: 1e4 x 60, one kppmix + tridiagonal pass per launch (mckpp_hip_vmix_pass),
                       with its own algorithmic bytes per column-pass
  tail                 24 steps of the settled headline run, a launch per step with the pass counts of each (one step
                       in seven has a column at itermax: the data-dependent tail), and the same steps as one launch
  drop_in              the reference-shaped host loop through the C-ABI, per step: mckpp_hip_set_forcing +
                       mckpp_hip_step + mckpp_hip_download of the scalar group / the restart set / every
                       field (PCIe-inclusive; never `value`)
  diurnal              the same columns through mckpp_hip_run_forced for 48 hourly steps of the
                       SURVEY 8(d) diurnal short-wave cycle (pass counts vary, kbl moves), GPU and CPU port
  other_shapes         1e5 x 69 levels on the stretched grid with 35 % land at dto = 1200 s (configs[4] shape)
                       and 1e5 x 100 levels (configs[3] shape), each with its own roofline fraction
  config3_long         configs[3]'s shape as a long run: 1e5 x 100, 300 steps in one call after 60 (columns at itermax in
                       every step: a run of many steps waits for their chains), with a census of the 24 steps after it
                       (a launch per step, pass counts); config3_long_12500: one GPU's share of it on 8 GPUs;
                       config3_long_two_ended_solver: the same with the opt-in solver
  strong_scaling_proxy the 12,500-column long run's rate relative to the 1e5-column long run's
"""
import argparse
import datetime
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

# The synthetic columns start from an analytic profile that is not in balance with the forcing: model
# step 2 then needs a mean of 34 vmix+ocnint passes per column instead of 6.  Both legs (GPU and CPU
# baseline) first run SPINUP untimed model steps as part of building the workload, so the warmup and
# timed steps are ordinary ones whatever --warmup is.
SPINUP = 3
DIURNAL_STEPS = 48
# Untimed steps of the headline workload right before the timed region: a cold MI355X runs ~0.2 s of fp64 work at a
# clock it does not hold (the first 40 steps measure 3.0 ms each, 400 in a row 3.5 ms - r03), so K timed steps after
# a short warmup describe a burst, not the run.  `value` is therefore measured after SETTLE more steps (~1.2 s); the
# burst figure is reported beside it.
SETTLE = 400

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def alg_bytes_per_column_step(nz, diag):
    """8*(20*nzp1 + 24), + 16*nzp1*8 with diagnostics written."""
    nzp1 = nz + 1
    b = 8 * (20 * nzp1 + 24)
    if diag:
        b += 16 * nzp1 * 8
    return b


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def diurnal_series(ncol, nt_first, nsteps, dto, index=None):
    """Hourly flux records of the SURVEY 8(d) diurnal cycle (mckpp_f90_amd/synth.py: flux_series)."""
    import common as cm

    return cm.synth.flux_series(ncol, nt_first, nsteps, dto, "bench", index)


def alg_bytes_per_column_pass(nz):
    """SURVEY.md section 8(d), configs[1] (un-fused kppmix + tridiagonal kernels, per pass):
    kppmix 8*(9*nzp1+10) + tridiagonal solves 8*(14*nzp1+8)."""
    nzp1 = nz + 1
    return 8 * (9 * nzp1 + 10) + 8 * (14 * nzp1 + 8)


def cpu_baseline(ncol_total, nz, warmup, nsteps, stride, diurnal_stride, diurnal_nt0, dto, with_diurnal=True):
    """Oracle (CPU restatement, OpenMP over columns, dynamic schedule like the reference's driver loop)
    on every `stride`-th column of the same workload: the same `warmup` untimed steps, then three
    consecutive blocks of `nsteps` model steps on all host cores (median reported; the first block is
    the steps the GPU leg times), and one block on a single thread over every 16th of those columns.
    The oracle runs with the library's portable exp (exp_mode=1, "custom exp"), which is what makes it
    bit-comparable with the device; its cost is that of libm's exp to within noise (2 calls per pass)."""
    import common as cm
    from oracle import orc

    idx = np.arange(0, ncol_total, stride)
    n = len(idx)
    cores = host_cores()
    oc, ob = cm.make_oracle(n, nz, mix="bench", exp_mode=1, index=idx, ntotal=ncol_total)
    warmup = warmup + SPINUP          # spin-up steps of the workload, then the same untimed warmup steps
    for nt in range(1, warmup + 1):
        orc.physics_driver(oc, ob, nt, nthreads=cores)
    times = []
    nt = warmup
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(nsteps):
            nt += 1
            orc.physics_driver(oc, ob, nt, nthreads=cores)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    passes = float(ob["npasses"].mean())
    # single thread, on a 16x thinner sample
    idx1 = idx[::16]
    oc1, ob1 = cm.make_oracle(len(idx1), nz, mix="bench", exp_mode=1, index=idx1, ntotal=ncol_total)
    for k in range(1, warmup + 1):
        orc.physics_driver(oc1, ob1, k, nthreads=1)
    t0 = time.perf_counter()
    for k in range(warmup + 1, warmup + nsteps + 1):
        orc.physics_driver(oc1, ob1, k, nthreads=1)
    dt1 = time.perf_counter() - t0
    out = {
        "value": n * nsteps / dt, "unit": "column-steps/s", "cores": cores, "kind": "port",
        "what": "C port of the reference (the oracle, oracle/mckpp_oracle.c), OpenMP over columns - not the reference's Fortran build",
        "exp": "library's portable exp (oracle exp_mode=1), not libm",
        "sample": (f"all {n} columns" if stride == 1 else f"every {stride}th column ({n} of {ncol_total})")
                  + f" of the workload, three blocks of {nsteps} model "
                  f"steps from step {warmup + 1} on (median {dt:.1f} s; all three: "
                  + ", ".join(f"{t:.1f}" for t in times) + f" s), {cores} OpenMP threads, dynamic schedule",
        "value_1thread": len(idx1) * nsteps / dt1,
        "sample_1thread": f"{len(idx1)} columns, model steps {warmup + 1}-{warmup + nsteps}, {dt1:.1f} s",
        "mean_passes_per_column_step_last_step": passes,
    }
    if not with_diurnal:
        return out
    # the diurnal leg on the CPU port: every diurnal_stride-th column, the same model steps as the GPU leg
    idxd = np.arange(0, ncol_total, diurnal_stride)
    nd = len(idxd)
    ocd, obd = cm.make_oracle(nd, nz, mix="bench", exp_mode=1, index=idxd, ntotal=ncol_total, dto=dto)
    nt0 = diurnal_nt0
    for k in range(1, nt0 + 1):
        orc.physics_driver(ocd, obd, k, nthreads=cores)
    series = diurnal_series(nd, nt0 + 1, DIURNAL_STEPS, dto, index=idxd)
    names = ("taux", "tauy", "swf", "lwf", "lhf", "shf", "rain", "snow")
    psum, pmax = 0.0, 0
    t0 = time.perf_counter()
    for r in range(DIURNAL_STEPS):
        orc.fluxes(ocd, obd, nt0 + 1 + r, **dict(zip(names, series[r])))
        orc.physics_driver(ocd, obd, nt0 + 1 + r, nthreads=cores)
        psum += float(obd["npasses"].mean())
        pmax = max(pmax, int(obd["npasses"].max()))
    dtd = time.perf_counter() - t0
    out["diurnal"] = {
        "value": nd * DIURNAL_STEPS / dtd, "unit": "column-steps/s", "cores": cores,
        "sample": f"every {diurnal_stride}th column ({nd}), model steps {nt0 + 1}-{nt0 + DIURNAL_STEPS}, {dtd:.1f} s",
        "mean_passes_per_column_step": psum / DIURNAL_STEPS, "max_passes": pmax,
    }
    return out


def time_steps(ctx, nt_first, nsteps, barrier):
    barrier()
    t0 = time.perf_counter()
    ctx.step(nt_first, nsteps)
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kms, nlaunch = ctx.last_kernel_ms()   # (nlaunch: the steps the call covered; step(nt, n) is one launch for all n)
    return dt, (kms / max(nlaunch, 1)) * 1e-3


def side_shape(mk, cm, ncol, nz, grid, dto, land_frac, steps, warmup, diag, dev_index):
    """One more workload shape on the same GPU: build, init, spin up, time `steps` model steps."""
    kc, k3 = cm.make_hip_case(ncol, nz, grid=grid, dto=dto)
    if land_frac > 0:
        land = (np.arange(ncol) * 7) % 20 < int(round(20 * land_frac))
        k3.run_physics[land] = 0
        k3.l_ocean[land] = 0
    ctx = mk.MckppHip(kc, device=dev_index)
    ctx.upload(k3)
    ctx.set_diagnostics(diag)
    ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench"))
    ctx.set_forcing(k3.sflux)
    ctx.step(1, SPINUP + warmup)
    ctx.synchronize()
    dt, kern_s = time_steps(ctx, 1 + SPINUP + warmup, steps, lambda: None)
    st, nflag, npass = ctx.status()
    nocean = int(ctx.ncolumns)
    balg = alg_bytes_per_column_step(nz, diag)
    achieved = balg * nocean / kern_s / 1e9
    out = {
        "workload": f"{ncol} grid points ({nocean} ocean columns) x {nz} levels, {grid} grid, dto={dto:.0f} s",
        "value": nocean * steps / dt, "unit": "column-steps/s", "ms_per_step": dt / steps * 1e3,
        "kernel": ctx.kernel_name, "kernel_avg_ms": kern_s * 1e3,
        "roofline_frac": achieved / HBM_PEAK_GBS, "achieved_GBps": achieved,
        "mean_passes_per_column_step_last_step": float(npass[k3.run_physics != 0].mean()),
        "flagged_columns_last_step": int(nflag),
    }
    ctx.close()
    del ctx, k3, kc
    gc.collect()
    return out


def long_run(mk, cm, ncol, nz, ntotal, dev_index, diag, settle=60, steps=300, census=24, solver_mode=0):
    """BASELINE configs[3]'s shape as a long run sees it: `ncol` of the `ntotal` closed-form columns (every
    ntotal/ncol-th: a GPU's round-robin share) x nz levels, `settle` untimed steps (past model step 60, from where
    columns iterate to itermax in every step), then `steps` model steps in ONE call - ms per step of that call -
    and a census of the `census` steps after it, a launch per step with the pass counts read after each: how many
    columns are at itermax per step, in what share of the steps, and what such a step takes when the host
    steps one call at a time."""
    stride = max(1, ntotal // ncol)
    idx = np.arange(0, ntotal, stride)[:ncol]
    kc, k3 = cm.make_hip_case(len(idx), nz, index=idx, ntotal=ntotal)
    ctx = mk.MckppHip(kc, device=dev_index)
    ctx.set_solver_mode(solver_mode)
    ctx.upload(k3)
    ctx.set_diagnostics(diag)
    ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(len(idx), "bench", index=idx))
    ctx.set_forcing(k3.sflux)
    ctx.step(1, settle)
    ctx.synchronize()
    nocean = int(ctx.ncolumns)
    dt, kern_s = time_steps(ctx, settle + 1, steps, lambda: None)
    nt = settle + steps + 1
    per_step = []
    for _ in range(census):
        d1, _k = time_steps(ctx, nt, 1, lambda: None)
        nt += 1
        st, nflag, npass = ctx.status()
        per_step.append((d1 * 1e3, int(npass.max()), int((npass > 50).sum()), int((npass > 12).sum()), float(npass.mean())))
    ctx.close()
    del ctx, k3, kc
    gc.collect()
    ms = np.array([q[0] for q in per_step]); mx = np.array([q[1] for q in per_step]); n50 = np.array([q[2] for q in per_step])
    balg = alg_bytes_per_column_step(nz, diag)
    return {
        "workload": f"{nocean} columns (every {stride}th of {ntotal}) x {nz} levels, bench forcing mix, model steps "
                    f"{settle + 1}-{settle + steps} in one call (one launch) after {settle} settle steps",
        "value": nocean * steps / dt, "unit": "column-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
        "kernel_avg_ms": kern_s * 1e3, "roofline_frac": balg * nocean / kern_s / 1e9 / HBM_PEAK_GBS,
        "census": {
            "what": f"the {census} model steps after the call, a launch per step (status read after each, outside the timing)",
            "ms_per_step_mean": float(ms.mean()), "ms_per_step_min": float(ms.min()), "ms_per_step_max": float(ms.max()),
            "max_passes": int(mx.max()), "share_of_steps_with_a_column_at_itermax": float((mx >= 200).mean()),
            "columns_over_50_passes_per_step": {"mean": float(n50.mean()), "min": int(n50.min()), "max": int(n50.max())},
            "columns_over_12_passes_per_step_mean": float(np.mean([q[3] for q in per_step])),
            "mean_passes_per_column_step": float(np.mean([q[4] for q in per_step])),
        },
    }


def config3_n1_reference():
    """The N = 1 figure config3_strong is to be compared with: config3_long of the last recorded N = 1 bench line
    (profiles/config3_long_n1.json, written from it with the library's build id)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "config3_long_n1.json")))
    except Exception as e:   # noqa: BLE001
        return {"error": f"profiles/config3_long_n1.json: {type(e).__name__}: {e}"}


def config3_strong_block(mk, cm, sharding, dist, rank, world, dev_index, diag, coll_dev, backend, barrier,
                         ntotal=100000, nz=100, settle=60, steps=200):
    """BASELINE configs[3] on the N ranks of this run: `ntotal` x `nz` columns dealt round-robin (strong scaling),
    `settle` untimed model steps, then `steps` more in ONE call per rank between barriers.  ms per step of the slowest
    rank, every rank's own time, and the N = 1 figure it is to be compared with.  A failure on any rank - or of a
    collective - becomes the block's text on every rank (sharding.guarded_block); it cannot take the line down."""
    import torch

    state = {}

    def _local():
        if os.environ.get("MCKPP_BENCH_FAIL_RANK3") == str(rank):   # (tests)
            raise RuntimeError("MCKPP_BENCH_FAIL_RANK3")
        idx = sharding.shard_indices(ntotal, rank, world)
        kc, k3 = cm.make_hip_case(len(idx), nz, index=idx, ntotal=ntotal)
        ctx = mk.MckppHip(kc, device=dev_index)
        ctx.upload(k3)
        ctx.set_diagnostics(diag)
        ctx.init_ocean(0)
        cm.set_forcing_3d(k3, cm.synth.forcing(len(idx), "bench", index=idx))
        ctx.set_forcing(k3.sflux)
        ctx.step(1, settle)
        ctx.synchronize()
        state["ctx"], state["n"] = ctx, int(ctx.ncolumns)
        return ctx

    def _timed(ctx):
        barrier()
        t0 = time.perf_counter()
        ctx.step(settle + 1, steps)
        ctx.synchronize()
        mine_s = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        st, nflag, npass = ctx.status()
        cdev = coll_dev if coll_dev is not None else "cpu"
        v = torch.tensor([mine_s, dt, float(state["n"]), float(npass.max()), float((npass > 50).sum())], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(every, v)
        rows = [[float(x) for x in e.tolist()] for e in every]
        tmax = max(r[1] for r in rows)
        ncols = int(sum(r[2] for r in rows))
        return {
            "workload": f"{ntotal} columns x {nz} levels in all, dealt round-robin over {world} ranks ({state['n']} on rank 0), model steps "
                        f"{settle + 1}-{settle + steps} in one call per rank after {settle} settle steps (BASELINE configs[3], strong scaling)",
            "value": ncols * steps / tmax, "unit": "column-steps/s", "ms_per_step": tmax / steps * 1e3,
            "per_rank_ms_per_step": [r[0] / steps * 1e3 for r in rows],
            "columns_over_50_passes_last_step_per_rank": [int(r[4]) for r in rows], "max_passes_last_step": int(max(r[3] for r in rows)),
            "backend": backend, "ranks": world,
            "n1_reference": config3_n1_reference(),
            "note": "a column that is at itermax step after step runs its steps one after the other: its chain of ~201 passes "
                    "per step bounds every rank alike, so the time per step falls with N only down to that chain",
        }

    try:
        out, err = sharding.guarded_block(dist, _local, _timed, device=coll_dev)
    except sharding.AgreementError as e:
        print(f"bench.py rank {rank}: {e}", file=sys.stderr, flush=True)
        os._exit(3)
    try:
        if "ctx" in state:
            state["ctx"].close()
    except Exception:   # noqa: BLE001
        pass
    return out if err is None else {"error": err}


def headline_variant(mk, cm, ncol, nz, idx, ntotal, a, dev_index, solver_mode=0, tail_frac=0.0):
    """The headline workload once more in a context of its own: spin-up, warmup and settle steps as the headline,
    then a timed region - with another solver mode, or (tail_frac > 0) stepped one launch at a time with the pass
    counts of every step, so that the iteration's data-dependent tail is inside a timed region and visible."""
    kc, k3 = cm.make_hip_case(ncol, nz, grid=a.grid, dto=a.dto, index=idx, ntotal=ntotal)
    ctx = mk.MckppHip(kc, device=dev_index)
    ctx.set_solver_mode(solver_mode)
    ctx.upload(k3)
    ctx.set_diagnostics(a.diag)
    ctx.init_ocean(0)
    sf = cm.synth.forcing(ncol, "bench", index=idx)
    cm.set_forcing_3d(k3, sf)
    ctx.set_forcing(k3.sflux)
    nt = 1
    ctx.step(nt, SPINUP + a.warmup + a.settle)
    ctx.synchronize()
    nt += SPINUP + a.warmup + a.settle
    nocean = int(ctx.ncolumns)
    balg = alg_bytes_per_column_step(nz, a.diag)
    if tail_frac <= 0.0:
        dt, kern_s = time_steps(ctx, nt, a.steps, lambda: None)
        st, nflag, npass = ctx.status()
        out = {"value": nocean * a.steps / dt, "unit": "column-steps/s", "ms_per_step": dt / a.steps * 1e3, "steps": a.steps,
               "kernel_avg_ms": kern_s * 1e3, "roofline_frac": balg * nocean / kern_s / 1e9 / HBM_PEAK_GBS,
               "mean_passes_per_column_step_last_step": float(npass.mean()), "flagged_columns_last_step": int(nflag)}
    else:
        # The data-dependent tail of the iteration, as the workload itself produces it: in a long run one step in seven
        # has a column that iterates to itermax (200 passes where the others take 6).  24 steps, first a launch per
        # step with the pass counts read after each (what a host that touches the state between steps gets), then -
        # in a second context brought to the same state - the same 24 steps as ONE launch (mckpp_hip_step(nt, 24)).
        # (Knocking columns out of balance does not produce a tail: 1 % of the columns with their old time level
        # pushed 3 m/s and 5 K away from the new one and storm forcing on top converged in 8.4 passes on average,
        # 12 at most - measured, r04.)
        nsingle = 24
        per_step, tsum, ksum = [], 0.0, 0.0
        for _ in range(nsingle):
            dt, kern_s = time_steps(ctx, nt, 1, lambda: None)
            nt += 1
            st, nflag, npass = ctx.status()
            tsum += dt; ksum += kern_s
            per_step.append({"ms": round(dt * 1e3, 3), "mean_passes": float(npass.mean()), "max_passes": int(npass.max()),
                             "columns_over_12_passes": int((npass > 12).sum()), "trap_fired": int(((st & 4) != 0).sum())})
        ctx.close()
        kc, k3 = cm.make_hip_case(ncol, nz, grid=a.grid, dto=a.dto, index=idx, ntotal=ntotal)
        ctx = mk.MckppHip(kc, device=dev_index)
        ctx.upload(k3)
        ctx.set_diagnostics(a.diag)
        ctx.init_ocean(0)
        cm.set_forcing_3d(k3, sf)
        ctx.set_forcing(k3.sflux)
        ctx.step(1, SPINUP + a.warmup + a.settle)
        ctx.synchronize()
        dt1, kern1 = time_steps(ctx, 1 + SPINUP + a.warmup + a.settle, nsingle, lambda: None)
        slow = [p for p in per_step if p["max_passes"] > 50]
        out = {"value": nocean * nsingle / tsum, "unit": "column-steps/s", "ms_per_step": tsum / nsingle * 1e3, "steps": nsingle,
               "kernel_avg_ms": ksum / nsingle * 1e3,
               "column_passes_per_s": sum(p["mean_passes"] for p in per_step) * nocean / tsum,
               "steps_with_a_column_over_50_passes": len(slow),
               "ms_of_those_steps": [p["ms"] for p in slow],
               "ms_of_the_other_steps_mean": float(np.mean([p["ms"] for p in per_step if p["max_passes"] <= 50])),
               "max_passes": max(p["max_passes"] for p in per_step),
               "the_same_steps_as_one_launch": {"value": nocean * nsingle / dt1, "ms_per_step": dt1 / nsingle * 1e3,
                                                "kernel_avg_ms": kern1 * 1e3},
               "per_step": per_step,
               "what": f"model steps {nt - nsingle}-{nt - 1} of the headline run, a launch per step (status read after each, outside "
                       "the timing), and again in a second context as one launch of all of them"}
    ctx.close()
    del ctx, k3, kc
    gc.collect()
    return out


def config1_pass(mk, cm, dev_index, a, ncol=10000, nz=60, launches=200):
    """BASELINE configs[1]: 1e4 columns x 60 levels, kppmix + tridiagonal solves only - one vmix + ocnint pass per
    launch (mckpp_hip_vmix_pass, kernel mode PASS: no iteration, no relaxation), from a spun-up state."""
    kc, k3 = cm.make_hip_case(ncol, nz)
    ctx = mk.MckppHip(kc, device=dev_index)
    ctx.upload(k3)
    ctx.set_diagnostics(a.diag)
    ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench"))
    ctx.set_forcing(k3.sflux)
    ctx.step(1, SPINUP + 5)
    for _ in range(20):
        ctx.vmix_pass(SPINUP + 6)
    ctx.synchronize()
    t0 = time.perf_counter()
    ksum = 0.0
    for _ in range(launches):
        ctx.vmix_pass(SPINUP + 6)
        ctx.synchronize()
        kms, nl = ctx.last_kernel_ms()
        ksum += kms / max(nl, 1)
    dt = time.perf_counter() - t0
    kern_s = ksum / launches * 1e-3
    nocean = int(ctx.ncolumns)
    bpass = alg_bytes_per_column_pass(nz)
    ach = bpass * nocean / kern_s / 1e9
    out = {"workload": f"{ncol} columns x {nz} levels, one kppmix + tridiagonal pass per launch (BASELINE configs[1])",
           "value": nocean / kern_s, "unit": "column-passes/s", "launches": launches,
           "kernel": ctx.kernel_name + ", mode PASS", "kernel_avg_ms": kern_s * 1e3, "wall_ms_per_launch_host_synchronous": dt / launches * 1e3,
           "algorithmic_bytes_per_column_pass": bpass,
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS},
           "note": "1e4 columns are 1.3 rounds of the 7,680 column slots in flight: the launch is mostly its own tail"}
    ctx.close()
    del ctx, k3, kc
    gc.collect()
    return out


def committed_counters(ncol, nz, kernel_name, build_id, solver_mode=0):
    """PMC-derived figures committed under profiles/ (HBM traffic per launch, VALU instruction counts).
    They describe one kernel build: returned only when recorded for the loaded library's build id and
    this workload, otherwise None with the reason."""
    tf = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(tf):
        return None, None, "profiles/counters.json not present"
    try:
        cj = json.load(open(tf))
    except Exception as e:   # noqa: BLE001
        return None, None, f"profiles/counters.json unreadable: {e}"
    if cj.get("build_id") != build_id:
        return None, None, (f"profiles/counters.json was recorded for build {cj.get('build_id')}, the loaded "
                            f"library is build {build_id}: not reported")
    for rec in cj.get("workloads", []):
        if (rec.get("ncol") == ncol and rec.get("nz") == nz and rec.get("kernel") == kernel_name
                and rec.get("solver_mode", 0) == solver_mode and "shape" not in rec):
            issue = None
            if "SQ_INSTS_VALU" in rec:
                passes = rec.get("passes_per_column", 6.0)
                issue = {
                    "source": f"profiles/counters.json ({cj.get('collected', 'rocprofv3 --pmc')}), build {build_id}",
                    "valu_wave_instructions_per_launch": rec["SQ_INSTS_VALU"],
                    "valu_instructions_per_column_pass": rec["SQ_INSTS_VALU"] / (ncol * passes),
                }
                if "SQ_ACTIVE_INST_VALU" in rec and "SQ_ACTIVE_INST_ANY" in rec:
                    issue["valu_active_over_wave_busy"] = rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_ACTIVE_INST_ANY"]
                if "SQ_LDS_BANK_CONFLICT" in rec:
                    issue["lds_bank_conflict_cycles"] = rec["SQ_LDS_BANK_CONFLICT"]
            return rec.get("hbm_bytes_per_launch"), issue, None
    return None, None, "no record for this workload in profiles/counters.json"


def launch_ranks(ngpus, argv, dry_run=False):
    """`bench.py --gpus N` started plainly: run the N ranks under torch.distributed.run as a child process and relay
    rank 0's JSON line and the exit code.  Nothing here imports torch or loads the library - a process that has
    initialised the GPU must not be the one that starts the ranks."""
    # --standalone: torchrun picks the rendezvous port itself (no bind-close-reuse race); 127.0.0.1 because the
    # container's host name may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={ngpus}", os.path.abspath(__file__)] + list(argv)
    if dry_run:
        print(json.dumps({"cmd": cmd, "torch_imported": "torch" in sys.modules,
                          "library_loaded": "mckpp_f90_amd" in sys.modules}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if r.returncode == 0 and line is None:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        return 1
    return r.returncode


def drop_in_block(mk, ctx, k3, nt0, nocean, masks):
    """The reference-shaped host loop on the C-ABI, one model step at a time: new forcing up
    (mckpp_hip_set_forcing: 6 doubles per column), mckpp_hip_step, the field groups of `mask` back into the
    Fortran-ordered host arrays (mckpp_hip_download)."""
    out = {}
    nt = nt0
    for name, mask, steps in masks:
        nt += 1                      # one untimed iteration: the first use of an array pins it (hipHostRegister)
        ctx.set_forcing(k3.sflux)
        ctx.step(nt, 1)
        ctx.download(k3, mask)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            nt += 1
            ctx.set_forcing(k3.sflux)
            ctx.step(nt, 1)
            ctx.download(k3, mask)
        dt = time.perf_counter() - t0
        out[name] = {"value": nocean * steps / dt, "unit": "column-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps}
    return out, nt


def single_process_block(mk, cm, sharding, kc_args, ncol_per_gpu, world, devices, steps, warmup, diag, strong_total):
    """The same shards behind one handle of the C-ABI (mckpp_hip_multi_*): all GPUs driven from this process."""
    import psutil

    ntotal = strong_total if strong_total else ncol_per_gpu * world
    note = None
    # bounded: this block runs on rank 0 while the other ranks wait, inside the driver's time limit for the whole
    # line - at 8 GPUs the ranks' 8e5 columns would be 23 GB of host arrays and ~25 s of uploads through one process
    cap = 200000
    if ntotal > cap:
        ntotal = cap
        note = f"{cap} columns in all ({cap // world} per GPU) instead of the ranks' total: bounded run time"
    need = ntotal * (kc_args["nz"] + 1) * 8 * 60          # host arrays of Kpp3dFields, generously
    avail = psutil.virtual_memory().available
    if need > 0.4 * avail:
        scale = max(1, int(need / (0.4 * avail)) + 1)
        ntotal = max(world, ntotal // scale)
        note = f"host memory: {ntotal} columns in all instead of the ranks' total"
    kc, k3 = cm.make_hip_case(ntotal, kc_args["nz"], grid=kc_args["grid"], dto=kc_args["dto"])
    m = mk.MckppHipMulti(kc, devices)
    m.upload(k3)
    m.set_diagnostics(diag)
    m.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ntotal, "bench"))
    m.set_forcing(k3.sflux)
    m.step(1, SPINUP + warmup)
    m.synchronize()
    t0 = time.perf_counter()
    m.step(1 + SPINUP + warmup, steps)
    m.synchronize()
    dt = time.perf_counter() - t0
    hm = np.zeros(ntotal, order="F")
    T = np.zeros((ntotal, kc_args["nz"] + 1), order="F")
    m.gather(4, 0, hm)          # first use pins the arrays and builds the root's buffers; not timed
    m.gather(2, 0, T)
    t0 = time.perf_counter()
    m.gather(4, 0, hm)
    t_h = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.gather(2, 0, T)
    t_T = time.perf_counter() - t0
    ok = bool(np.isfinite(hm).all() and (hm > 0).all() and np.isfinite(T).all())
    out = {
        "what": "mckpp_hip_multi_step on all shards from one process (rank 0's), then mckpp_hip_multi_gather of hmix and T "
                "to shard 0 (peer copies over the GPU interconnect, all in flight at once) and to the host",
        "devices": [int(d) for d in devices], "columns": int(ntotal), "value": ntotal * steps / dt, "unit": "column-steps/s",
        "ms_per_step": dt / steps * 1e3, "gather_hmix_ms": t_h * 1e3, "gather_T_ms": t_T * 1e3,
        "gather_T_GBps_to_host": T.nbytes / t_T / 1e9, "finite": ok,
    }
    if note:
        out["note"] = note
    m.close()
    return out


def run_single_process_leg(a, devs):
    """The one-handle leg in a child process of rank 0: its peer copies and cross-device waits have never run on more
    than one physical device (no multi-GPU node was available to this build), and whatever happens to it - an
    exception, a crash of the process, a hang - the ranks' own line must still be printed."""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                        "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--single-process-leg", ",".join(str(d) for d in devs),
           "--steps", str(a.steps), "--warmup", str(a.warmup), "--ncol", str(a.ncol), "--nz", str(a.nz),
           "--grid", a.grid, "--dto", str(a.dto), "--diag", str(a.diag), "--total-ncol", str(a.total_ncol)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "the child process did not finish within 240 s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"child process exit code {r.returncode}: {(r.stderr or r.stdout).strip()[-400:]}"}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": f"unreadable result of the child process: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncol", type=int, default=100000, help="columns per GPU")
    ap.add_argument("--nz", type=int, default=60)
    ap.add_argument("--diag", type=int, default=1, help="write the per-step diagnostic fields (reference behaviour)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip diurnal / other_shapes / strong_scaling_proxy")
    ap.add_argument("--cpu-stride", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=10, help="model steps per timed block of the CPU baseline")
    ap.add_argument("--grid", default="uniform")
    ap.add_argument("--dto", type=float, default=3600.0)
    ap.add_argument("--land", type=float, default=0.0, help="fraction of land points (run_physics = .F.)")
    ap.add_argument("--total-ncol", type=int, default=0,
                    help="strong scaling: this many columns in all, divided over the GPUs (overrides --ncol)")
    ap.add_argument("--sustained-steps", type=int, default=400)
    ap.add_argument("--settle", type=int, default=SETTLE,
                    help="untimed steps of the same work before the timed region (0: time the cold device's burst)")
    ap.add_argument("--legs", default="all", help="comma list of the extra legs to run (two_ended_solver, tail, config1_pass, ...): "
                                                 "for experiments; default all")
    ap.add_argument("--print-launch", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-process-leg", default="", help=argparse.SUPPRESS)   # internal: see run_single_process_leg
    a = ap.parse_args()

    if a.single_process_leg:   # child of rank 0: the one-handle leg alone, its JSON on stdout
        import common as cm
        import mckpp_f90_amd as mk
        from mckpp_f90_amd import sharding

        devs = [int(x) for x in a.single_process_leg.split(",")]
        print(json.dumps(single_process_block(mk, cm, sharding, {"nz": a.nz, "grid": a.grid, "dto": a.dto}, a.ncol,
                                              len(devs), devs, a.steps, a.warmup, a.diag, a.total_ncol)), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:   # started plainly: become the launcher of the N ranks
        argv = [x for x in sys.argv[1:] if x != "--print-launch"]
        raise SystemExit(launch_ranks(a.gpus, argv, dry_run=a.print_launch))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {a.gpus}")

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    dist = None
    # Rehearsal knobs (never set by the driver): MCKPP_BENCH_BACKEND=gloo runs the collectives on
    # the CPU and MCKPP_BENCH_SHARE_GPU=1 puts every rank on device 0, so the N>1 code path can be
    # exercised on a one-GPU box.
    backend = os.environ.get("MCKPP_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("MCKPP_BENCH_SHARE_GPU") else local_rank
    coll_dev = torch.device("cuda", dev_index) if backend == "nccl" else None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index),
                                    timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))

    import common as cm
    import mckpp_f90_amd as mk
    from mckpp_f90_amd import api as mkapi

    from mckpp_f90_amd import sharding

    nz = a.nz
    strong = a.total_ncol > 0
    ntotal = a.total_ncol if strong else a.ncol * world
    idx = sharding.shard_indices(ntotal, rank, world)   # round-robin shard of one global closed-form set
    ncol = len(idx)
    kc, k3 = cm.make_hip_case(ncol, nz, grid=a.grid, dto=a.dto, index=idx, ntotal=ntotal)
    if a.land > 0:
        land = (np.arange(ncol) * 7) % 20 < int(round(20 * a.land))
        k3.run_physics[land] = 0
        k3.l_ocean[land] = 0
    ctx = mk.MckppHip(kc, device=dev_index)
    ctx.upload(k3)
    ctx.set_diagnostics(a.diag)
    ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench", index=idx))
    ctx.set_forcing(k3.sflux)
    ctx.synchronize()
    nocean = int(ctx.ncolumns)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.step(1, SPINUP)
    if a.warmup > 0:
        ctx.step(1 + SPINUP, a.warmup)
    nt_next = 1 + SPINUP + a.warmup
    burst = None
    if a.settle > 0:   # the K steps as a cold device runs them (reported, not `value`), then the settle leg
        dtb, kern_b = time_steps(ctx, nt_next, a.steps, barrier)
        nt_next += a.steps
        burst = {"value_this_rank": nocean * a.steps / dtb, "unit": "column-steps/s", "ms_per_step": dtb / a.steps * 1e3,
                 "kernel_avg_ms": kern_b * 1e3,
                 "what": f"the same {a.steps} steps right after the {a.warmup} warmup steps, device cold (this rank's columns)"}
        ctx.step(nt_next, a.settle)
        ctx.synchronize()
        nt_next += a.settle
    dt, kern_s = time_steps(ctx, nt_next, a.steps, barrier)
    nt_next += a.steps

    st, nflag, npass = ctx.status()
    ocean = k3.run_physics != 0
    ncols_all = nocean
    multi = None
    if dist is not None:
        cdev = coll_dev if coll_dev is not None else "cpu"
        mine = torch.tensor([dt], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) / a.steps * 1e3 for x in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([float(nocean)], dtype=torch.float64, device=cdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        ncols_all = int(c.item())
        ones = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)     # ranks that took part in a collective on `cdev` tensors
        multi = {
            "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "backend": dist.get_backend(), "rccl_ranks": int(round(float(ones.item()))) if backend == "nccl" else None,
            "collective_ranks": int(round(float(ones.item()))), "world_size": dist.get_world_size(),
        }
        # diagnostics gather (not timed into `value`): hmix and T of every rank's columns to rank 0.  The line does not
        # depend on it - but every rank must take the same way through it: a rank whose download fails must not leave
        # the others inside a gather (sharding.guarded_block: the ranks agree on failures before and after the
        # collectives; if they cannot, the job ends with a non-zero code instead of hanging).
        def _local_download():
            if os.environ.get("MCKPP_BENCH_FAIL_RANK") == str(rank):   # (tests: a failure on one rank alone)
                raise RuntimeError("MCKPP_BENCH_FAIL_RANK")
            t0 = time.perf_counter()
            ctx.download(k3, mk.api.F_SCALARS | mk.api.F_PROFILES)
            return time.perf_counter() - t0

        def _gathers(t_down):
            barrier()
            t0 = time.perf_counter()
            parts = sharding.gather_to_root(k3.hmix, dist, device=coll_dev)
            if coll_dev is not None:
                torch.cuda.synchronize()
            t_h = time.perf_counter() - t0
            barrier()
            t0 = time.perf_counter()
            tparts = sharding.gather_to_root(np.ascontiguousarray(k3.X[:, :, 0]), dist, device=coll_dev)
            if coll_dev is not None:
                torch.cuda.synchronize()
            t_T = time.perf_counter() - t0
            g = {"what": "hmix (8 B/column) and T (8 (nz+1) B/column) of every rank to rank 0, torch.distributed.gather on "
                         + ("device tensors (RCCL over xGMI)" if backend == "nccl" else "host tensors (gloo rehearsal)"),
                 "download_scalars_and_profiles_ms": t_down * 1e3, "hmix_ms": t_h * 1e3, "T_ms": t_T * 1e3,
                 "T_bytes_per_rank": int(ncol * (nz + 1) * 8)}
            if rank == 0:
                hmix_all = sharding.unshard(parts, ntotal)
                T_all = sharding.unshard(tparts, ntotal)
                g["complete_and_finite"] = bool(np.isfinite(hmix_all).all() and hmix_all.shape == (ntotal,)
                                                and np.isfinite(T_all).all() and T_all.shape == (ntotal, nz + 1))
                g["checked"] = ("hmix and T of all ranks finite and complete on rank 0" if g["complete_and_finite"] else
                                "FAILED: hmix / T gathered on rank 0 incomplete or not finite")
            return g

        try:
            g, gerr = sharding.guarded_block(dist, _local_download, _gathers, device=coll_dev)
        except sharding.AgreementError as e:
            print(f"bench.py rank {rank}: {e}", file=sys.stderr, flush=True)
            os._exit(3)
        multi["gather"] = g if gerr is None else {"error": gerr}
        # BASELINE configs[3] itself, inside the one command the driver runs: 1e5 x 100 columns dealt round-robin over
        # the N ranks (strong scaling), model steps 61-260 in ONE call per rank
        multi["config3_strong"] = config3_strong_block(mk, cm, sharding, dist, rank, world, dev_index, a.diag, coll_dev, backend, barrier)

    out = None
    if rank == 0:
        balg = alg_bytes_per_column_step(nz, a.diag)
        achieved = balg * nocean / kern_s / 1e9
        build = mkapi.build_id()
        traffic, issue, why_not = committed_counters(ncol, nz, ctx.kernel_name, build, ctx.solver_mode)
        out = {
            "metric": "column-steps/s at 1e5 cols x 60 levels, 1/2/4/8 GPU; % HBM roofline",
            "value": ncols_all * a.steps / dt,
            "unit": "column-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"{ntotal} columns x {nz} levels in all, {ncol} per GPU" if strong else
                             f"{ncol} columns x {nz} levels per GPU") + ", full ocnstep (kppmix stack + swfrac + "
                            "state equation + tridiagonal solves), bench forcing mix (1/3 stable, 1/3 convective, "
                            f"1/3 windy), dto={a.dto:.0f} s, "
                            + ("BASELINE configs[3] shape" if strong and nz == 100 else "BASELINE configs[2]"),
                "columns_per_gpu": ncol, "ocean_columns_per_gpu": nocean, "levels": nz,
                "diagnostics_written": bool(a.diag), "spin_up_steps": SPINUP, "settle_steps": a.settle,
                "solver": "reference order (mckpp_hip_set_solver_mode 0; solvers.F90:112-161)" if ctx.solver_mode == 0
                          else "two-ended elimination (mckpp_hip_set_solver_mode 1, opt-in)",
                "sharding": f"columns round-robin over {world} GPU(s), no data-path collective",
                "mean_passes_per_column_step_last_step": float(npass[ocean].mean()),
                "max_passes_last_step": int(npass[ocean].max()),
                "columns_over_12_passes_last_step": int((npass[ocean] > 12).sum()),
                "columns_over_50_passes_last_step": int((npass[ocean] > 50).sum()),
                "flagged_columns_last_step": int(nflag),
                "library_build": build,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_without_diagnostic_bytes": alg_bytes_per_column_step(nz, 0) * nocean / kern_s / 1e9 / HBM_PEAK_GBS,
                "kernel": ctx.kernel_name + " (cooperative, persistent)",
                "kernel_avg_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": balg * nocean,
                "kernel_launches_in_the_timed_region": ctx.last_launch_count(),
                "launch_note": ("kernel_avg_ms and algorithmic_bytes_per_launch are per model step; the timed region's "
                                f"{a.steps} steps are {ctx.last_launch_count()} launch(es) of k_column_ps (mckpp_hip_step(nt, n): one "
                                "launch takes every column through all n steps), so a kernel trace shows dispatches of "
                                f"{a.steps // max(ctx.last_launch_count(), 1)} x kernel_avg_ms"),
                "note": "fp64 instruction-issue / dependent-chain bound, not HBM-bound (DESIGN.md section 6)",
                "issue": issue,
            },
        }
        if why_not:
            out["roofline"]["counters_note"] = why_not
        out["column_passes_per_s"] = out["value"] * float(npass[ocean].mean())
        if burst is not None:
            out["burst"] = burst
        if world > 1:
            out["roofline"]["note_ranks"] = ("rank 0's kernel on its own columns; every rank launches the same kernel on "
                                             "its shard - per-rank step times in multi_gpu.per_rank_ms_per_step")
        if multi is not None:
            out["multi_gpu"] = multi
            # the state gathered on rank 0 must be complete and finite: anything else fails the run (transport errors of
            # the diagnostics alone do not)
            out["ok"] = not (isinstance(multi.get("gather"), dict) and multi["gather"].get("complete_and_finite") is False)

    if dist is not None:   # the same shards behind one handle, from rank 0's process; the other ranks wait
        barrier()
        if rank == 0:
            devs = [0] * world if os.environ.get("MCKPP_BENCH_SHARE_GPU") else list(range(world))
            out["multi_gpu"]["single_process"] = run_single_process_leg(a, devs)
        barrier()

    extras = world == 1 and not a.no_extras
    if extras and a.legs != "all":   # experiments: only the named legs of the second block below
        ctx.close()
        del ctx, k3, kc
        gc.collect()
        legs = a.legs.split(",")
        if "two_ended_solver" in legs:
            out["two_ended_solver"] = headline_variant(mk, cm, ncol, nz, idx, ntotal, a, dev_index, solver_mode=1)
        if "tail" in legs:
            out["tail"] = headline_variant(mk, cm, ncol, nz, idx, ntotal, a, dev_index, tail_frac=0.01)
        if "config1_pass" in legs:
            out["config1_pass"] = config1_pass(mk, cm, dev_index, a)
        if "config3_long" in legs:
            out["config3_long"] = long_run(mk, cm, 100000, 100, 100000, dev_index, a.diag)
        if "config3_long_12500" in legs:
            out["config3_long_12500"] = long_run(mk, cm, 12500, 100, 100000, dev_index, a.diag)
        if "config3_long_two_ended_solver" in legs:
            out["config3_long_two_ended_solver"] = long_run(mk, cm, 100000, 100, 100000, dev_index, a.diag, census=4, solver_mode=1)
        print(json.dumps(out), flush=True)
        return
    if extras:
        nt_s = nt_next - 1
        # ---- drop-in host loop: forcing up, one step, field groups down, every step ----
        out["drop_in"], nt_s = drop_in_block(mk, ctx, k3, nt_s, nocean, (
            ("scalars", mk.api.F_SCALARS, 20), ("restart_set", mk.api.F_RESTART, 6), ("all_fields", mk.api.F_ALL, 3)))
        out["drop_in"]["what"] = ("per model step mckpp_hip_set_forcing + mckpp_hip_step + mckpp_hip_download(mask) into "
                                  "Fortran-ordered host arrays (pinned on first use); PCIe-inclusive, never `value`")
        # ---- diurnal leg: the reference's forced time loop from resident flux records ----
        nt0 = nt_s
        series = diurnal_series(ncol, nt0 + 1, DIURNAL_STEPS, a.dto, index=idx)
        ctx.set_flux_series(nt0, series)
        del series
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run_forced(nt0 + 1, DIURNAL_STEPS, 1)
        ctx.synchronize()
        dtd = time.perf_counter() - t0
        kms, nl = ctx.last_kernel_ms()
        st, nflag, npass = ctx.status()
        out["diurnal"] = {
            "workload": f"same {nocean} columns, model steps {nt0 + 1}-{nt0 + DIURNAL_STEPS} through mckpp_hip_run_forced: "
                        "hourly flux records resident in HBM, mckpp_fluxes on the stream before every step, bench mix "
                        "with swf = max(0, 800 sin(2 pi t / 86400))",
            "value": nocean * DIURNAL_STEPS / dtd, "unit": "column-steps/s", "ms_per_step": dtd / DIURNAL_STEPS * 1e3,
            "stream_ms_per_step_incl_fluxes": kms / max(nl, 1),
            "mean_passes_per_column_step_last_step": float(npass[ocean].mean()),
            "max_passes_last_step": int(npass[ocean].max()), "flagged_columns_last_step": int(nflag),
        }
    ctx.close()
    del ctx, k3, kc
    gc.collect()

    if extras:
        out["two_ended_solver"] = headline_variant(mk, cm, ncol, nz, idx, ntotal, a, dev_index, solver_mode=1)
        out["two_ended_solver"]["config"] = {
            "solver": "two-ended elimination (mckpp_hip_set_solver_mode 1): opt-in, not the reference's order of operations; "
                      "bit-identical to the oracle's solver_mode=1, within rounding of mode 0 (profiles/r04/parity_tolerance.json)",
            "ratio_to_value": out["two_ended_solver"]["value"] / out["value"]}
        out["tail"] = headline_variant(mk, cm, ncol, nz, idx, ntotal, a, dev_index, tail_frac=0.01)
        out["config1_pass"] = config1_pass(mk, cm, dev_index, a)
        out["other_shapes"] = [
            side_shape(mk, cm, 100000, 69, "stretched", 1200.0, 0.35, 10, 2, a.diag, dev_index),
            side_shape(mk, cm, 100000, 100, "uniform", 3600.0, 0.0, 10, 2, a.diag, dev_index),
        ]
        # BASELINE configs[3] as a long run: past model step 60 some columns iterate to itermax in every step, and a
        # column's steps follow each other - its chain of 201 passes per step, not the device's throughput, is what a
        # run of many steps waits for.  1e5 columns on one GPU, and one GPU's share of them on 8 (every 8th column).
        out["config3_long"] = long_run(mk, cm, 100000, 100, 100000, dev_index, a.diag)
        out["config3_long_12500"] = long_run(mk, cm, 12500, 100, 100000, dev_index, a.diag)
        out["config3_long_two_ended_solver"] = long_run(mk, cm, 100000, 100, 100000, dev_index, a.diag, census=4, solver_mode=1)
        small60 = side_shape(mk, cm, 12500, 60, "uniform", 3600.0, 0.0, 20, 2, a.diag, dev_index)
        l1, l8 = out["config3_long"], out["config3_long_12500"]
        out["strong_scaling_proxy"] = {
            "what": "one GPU on 12,500 columns = its share of configs[3] (1e5 x 100 over 8 GPUs, every 8th column), 300 steps in "
                    "one call past model step 60: its rate relative to the same GPU's rate on all 1e5 columns bounds the 8-GPU "
                    "strong-scaling efficiency.  Bounded by the chain of a column that is at itermax step after step, not by "
                    "throughput (config3_long_12500.census)",
            "nz100": {"value": l8["value"], "ms_per_step": l8["ms_per_step"], "ratio_to_1e5": l8["value"] / l1["value"],
                      "ms_per_step_1e5": l1["ms_per_step"]},
            "nz60": {"value": small60["value"], "ms_per_step": small60["ms_per_step"],
                     "ratio_to_1e5": small60["value"] / out["value"],
                     "note": "20 steps right after the spin-up (round 4's window), throughput only"},
        }
    if extras and a.sustained_steps > 0:
        # ---- sustained: the headline workload again, many steps in one call.  Last of the GPU legs: after ~1 s of
        # continuous fp64 work the device settles on a lower clock, which would colour every leg run after it.
        kc, k3 = cm.make_hip_case(ncol, nz, grid=a.grid, dto=a.dto, index=idx, ntotal=ntotal)
        ctx = mk.MckppHip(kc, device=dev_index)
        ctx.upload(k3)
        ctx.set_diagnostics(a.diag)
        ctx.init_ocean(0)
        cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench", index=idx))
        ctx.set_forcing(k3.sflux)
        ctx.step(1, SPINUP + a.warmup)
        ctx.synchronize()
        dts, kern_ss = time_steps(ctx, 1 + SPINUP + a.warmup, a.sustained_steps, barrier)
        quarter = max(1, a.sustained_steps // 4)
        dtq, _ = time_steps(ctx, 1 + SPINUP + a.warmup + a.sustained_steps, quarter, barrier)
        out["sustained"] = {"steps": a.sustained_steps, "value": nocean * a.sustained_steps / dts, "unit": "column-steps/s",
                            "ms_per_step": dts / a.sustained_steps * 1e3, "kernel_avg_ms": kern_ss * 1e3,
                            "ms_per_step_of_the_next_%d_steps" % quarter: dtq / quarter * 1e3,
                            "note": "same workload and state as the headline, one call (one launch): a column that iterates to "
                                    "itermax delays its own next step only; the headline's 40-step window ends on such columns "
                                    "more often per step than a 400-step one (DESIGN.md section 6)"}
        # the drop-in loop once more, later in the run (a long model run's condition: the columns past their spin-up)
        again, _ = drop_in_block(mk, ctx, k3, 1 + SPINUP + a.warmup + a.sustained_steps + quarter, nocean,
                                 (("scalars", mk.api.F_SCALARS, 40), ("restart_set", mk.api.F_RESTART, 8)))
        out["drop_in"]["scalars_after_the_sustained_leg"] = again["scalars"]
        out["drop_in"]["restart_set_after_the_sustained_leg"] = again["restart_set"]
        ctx.close()
        del ctx, k3, kc
        gc.collect()
    if rank == 0:
        if world > 1 and not a.no_cpu_baseline:
            # rank 0's host cores, every 4th of rank 0's columns (bounded: the line must come within the driver's limit
            # at every N), no diurnal leg
            cb = cpu_baseline(ncol, nz, a.warmup, a.cpu_steps, max(a.cpu_stride, 4), 4, 0, a.dto, with_diurnal=False)
            cb["note"] = "timed on rank 0's host while the other ranks wait; a sample of rank 0's shard of the workload"
            out["cpu_baseline"] = cb
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(ncol, nz, a.warmup, a.cpu_steps, a.cpu_stride, 4, SPINUP + a.warmup + a.steps, a.dto)
            if extras:
                cb["diurnal"]["note"] = ("the CPU leg runs the diurnal steps right after the headline's; the GPU leg after its "
                                         "sustained and drop-in legs (later model steps of the same cycle)")
            if extras:
                out["diurnal"]["cpu_port"] = cb.pop("diurnal")
            else:
                cb.pop("diurnal", None)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if out is not None and out.get("ok") is False:
        raise SystemExit("bench.py: the state gathered on rank 0 is incomplete or not finite")


if __name__ == "__main__":
    main()
