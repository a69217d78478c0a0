#!/usr/bin/env python3
"""bench.py - column-steps/s of the MC-KPP column-physics step on MI355X.

Workload (BASELINE.json configs[2]): 1e5 synthetic columns x 60 levels per GPU
(spun up for SPINUP model steps, see below), full ocnstep (KPP mixing stack with swfrac + equation of state, tridiagonal
solves), fp64, state resident in HBM.  A "step" is one mckpp_physics_driver
call over the rank's columns.  Columns shard across GPUs with no data-path
collective (weak scaling: 1e5 columns per GPU); a torch.distributed (RCCL)
gather of hmix to rank 0 runs after the timed region only, as the diagnostics
gather the path has.

Prints ONE JSON line on rank 0 (contract in the task statement) with the two
extra objects `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
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


def cpu_baseline(ncol_total, nz, warmup, nsteps, stride):
    """Oracle (CPU restatement, OpenMP over columns, dynamic schedule like the reference's driver loop)
    on every `stride`-th column of the same workload: the same `warmup` untimed steps, then three
    consecutive blocks of `nsteps` model steps on all host cores (median reported; the first block is
    the steps the GPU leg times), and one block on a single thread over every 16th of those columns."""
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
    return {
        "value": n * nsteps / dt, "unit": "column-steps/s", "cores": cores, "kind": "port",
        "sample": (f"all {n} columns" if stride == 1 else f"every {stride}th column ({n} of {ncol_total})")
                  + f" of the workload, three blocks of {nsteps} model "
                  f"steps from step {warmup + 1} on (median {dt:.1f} s; all three: "
                  + ", ".join(f"{t:.1f}" for t in times) + f" s), {cores} OpenMP threads, dynamic schedule",
        "value_1thread": len(idx1) * nsteps / dt1,
        "sample_1thread": f"{len(idx1)} columns, model steps {warmup + 1}-{warmup + nsteps}, {dt1:.1f} s",
        "mean_passes_per_column_step_last_step": passes,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncol", type=int, default=100000, help="columns per GPU")
    ap.add_argument("--nz", type=int, default=60)
    ap.add_argument("--diag", type=int, default=1, help="write the per-step diagnostic fields (reference behaviour)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=1)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import common as cm
    import mckpp_f90_amd as mk

    ncol, nz = a.ncol, a.nz
    ntotal = ncol * world
    from mckpp_f90_amd import sharding

    idx = sharding.shard_indices(ntotal, rank, world)   # round-robin shard of one global closed-form set
    kc, k3 = cm.make_hip_case(ncol, nz, index=idx, ntotal=ntotal)
    ctx = mk.MckppHip(kc, device=dev_index)
    ctx.upload(k3)
    ctx.set_diagnostics(a.diag)
    ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench", index=idx))
    ctx.set_forcing(k3.sflux)
    ctx.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.step(1, SPINUP)
    if a.warmup > 0:
        ctx.step(1 + SPINUP, a.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.step(1 + SPINUP + a.warmup, a.steps)
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kms, nlaunch = ctx.last_kernel_ms()

    st, nflag, npass = ctx.status()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # diagnostics gather (not timed): hmix of every rank's columns to rank 0 over RCCL
        ctx.download(k3, mk.api.F_SCALARS)
        parts = sharding.gather_to_root(k3.hmix, dist, device=coll_dev)
        if rank == 0:
            hmix_all = sharding.unshard(parts, ntotal)
            assert np.isfinite(hmix_all).all() and hmix_all.shape == (ntotal,)

    if rank == 0:
        kern_s = (kms / max(nlaunch, 1)) * 1e-3
        balg = alg_bytes_per_column_step(nz, a.diag)
        achieved = balg * ncol / kern_s / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("ncol") == ncol and tj.get("nz") == nz:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # instruction-issue side of the story (the kernel is not HBM-bound): committed PMC counters of
        # the same workload, per launch; profiles/r01/README.md says how they were collected
        issue = None
        pf = os.path.join(ROOT, "profiles", "r01", "pmc_sq_final.json")
        if os.path.exists(pf) and ncol == 100000 and nz == 60:
            try:
                c = json.load(open(pf))["counters"]
                issue = {
                    "source": "profiles/r01/pmc_sq_final.json (rocprofv3 --pmc, same workload, 6 passes per column)",
                    "valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
                    "valu_instructions_per_column_pass": c["SQ_INSTS_VALU"] / (ncol * 6.0),
                    "valu_active_over_wave_busy": c["SQ_ACTIVE_INST_VALU"] / c["SQ_ACTIVE_INST_ANY"],
                    "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
                }
            except Exception:
                issue = None
        out = {
            "metric": "column-steps/s at 1e5 cols x 60 levels, 1/2/4/8 GPU; % HBM roofline",
            "value": ntotal * a.steps / dt,
            "unit": "column-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{ncol} columns x {nz} levels per GPU, full ocnstep (kppmix stack + swfrac + "
                            "state equation + tridiagonal solves), bench forcing mix (1/3 stable, 1/3 convective, "
                            "1/3 windy), dto=3600 s, BASELINE configs[2]",
                "columns_per_gpu": ncol, "levels": nz, "diagnostics_written": bool(a.diag),
                "spin_up_steps": SPINUP,
                "sharding": f"columns round-robin over {world} GPU(s), no data-path collective",
                "mean_passes_per_column_step_last_step": float(npass.mean()),
                "flagged_columns_last_step": int(nflag),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": ctx.kernel_name + " (cooperative, persistent)",
                "kernel_avg_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": balg * ncol,
                "note": "fp64 instruction-issue bound, not HBM-bound (DESIGN.md section 6)", "issue": issue,
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ncol, nz, a.warmup, a.steps, a.cpu_stride)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
