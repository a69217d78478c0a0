#!/usr/bin/env python3
"""Digest the rocprofv3 output of tools/r05_profile.sh: per-kernel stats of the default bench run, the
column-kernel rows of its trace, and the PMC counters of the LAST column-kernel dispatch of each counter run
(the third timed step: 6 passes per column) -> profiles-ready files."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
sys.path.insert(0, os.getcwd())


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


res = {"workloads": []}
try:
    from mckpp_f90_amd import api
    res["build_id"] = api.build_id()
except Exception as e:   # noqa: BLE001
    res["build_id"] = f"unknown ({e})"
res["collected"] = "rocprofv3 --kernel-trace --pmc <counters> (separate passes: two SQ groups, FETCH_SIZE, WRITE_SIZE) on bench.py --steps 3 --warmup 2 --settle 0 --no-extras --no-cpu-baseline (MCKPP_SOLVER_MODE per record), last column-kernel dispatch; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), KB x 1024"

for shape, nz, sm in (("60", 60, 0), ("60", 60, 1), ("69s", 69, 0), ("100", 100, 0), ("100", 100, 1)):
    t = f"{shape}_sm{sm}"
    rec = {"ncol": 100000, "nz": nz, "solver_mode": sm, "passes_per_column": 6.0}
    if shape.startswith("69s"):
        rec["shape"] = "stretched grid, 35 % land (65,000 ocean columns), dto 1200 s" + (
            "; MCKPP_PS_CONFLICT_FREE=1 (slot stride without LDS bank conflicts in the level-major phases: experiment)" if shape == "69sfree" else "")
    for tag in ("sq1", "sq2", "fetch", "write"):
        files = find(f"{tag}_{t}/**/*counter_collection.csv")
        if not files:
            continue
        rows = list(csv.DictReader(open(files[0])))
        col = [r for r in rows if "k_column" in r.get("Kernel_Name", "")]
        if not col:
            continue
        last = max(int(r["Dispatch_Id"]) for r in col)
        for r in col:
            if int(r["Dispatch_Id"]) == last:
                rec[r["Counter_Name"]] = float(r["Counter_Value"])
                name = r["Kernel_Name"]
                rec["kernel_symbol"] = name[:120]
    nsteps = 1
    try:
        j = json.loads(open(os.path.join(out, f"sq1_{t}.json")).read().strip().splitlines()[-1])
        # the timed region of the counter runs is ONE dispatch of `steps` model steps (mckpp_hip_step(nt, n)): the
        # counters of that dispatch are divided by its steps, so that every figure is per model step like round 3's
        nsteps = max(1, j["steps"] // max(1, j["roofline"].get("kernel_launches_in_the_timed_region", j["steps"])))
        rec["steps_in_the_dispatch"] = nsteps
        for k in list(rec):
            if k.startswith("SQ_") or k in ("FETCH_SIZE", "WRITE_SIZE"):
                rec[k] = rec[k] / nsteps
        rec["kernel"] = j["roofline"]["kernel"].split(" ")[0]
        rec["algorithmic_bytes_per_launch"] = j["roofline"]["algorithmic_bytes_per_launch"]
        rec["ocean_columns"] = j["config"]["ocean_columns_per_gpu"]
        rec["kernel_avg_ms_under_the_profiler"] = j["roofline"]["kernel_avg_ms"]
    except Exception:   # noqa: BLE001
        pass
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes_per_launch"] = int(2 * rec["FETCH_SIZE"] * 1024 + rec["WRITE_SIZE"] * 1024)
        if rec.get("algorithmic_bytes_per_launch"):
            rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    if "SQ_LDS_BANK_CONFLICT" in rec and rec.get("SQ_ACTIVE_INST_LDS"):
        rec["lds_bank_conflict_over_lds_active"] = rec["SQ_LDS_BANK_CONFLICT"] / rec["SQ_ACTIVE_INST_LDS"]
    if "SQ_WAIT_ANY" in rec and rec.get("SQ_WAVE_CYCLES"):
        rec["wait_any_over_wave_cycles"] = rec["SQ_WAIT_ANY"] / rec["SQ_WAVE_CYCLES"]
    res["workloads"].append(rec)
json.dump(res, open(os.path.join(out, "counters.json"), "w"), indent=1)

# kernel stats + the column-kernel rows of the trace of the default bench run
for f in find("stats/**/*kernel_stats.csv"):
    os.replace(f, os.path.join(out, "bench_default_kernel_stats.csv"))
for f in find("stats/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "k_column" in r.get("Kernel_Name", "")]
    with open(os.path.join(out, "bench_default_kernel_trace_column_kernels.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Duration_ns", "Workgroup_Size", "Grid_Size", "LDS_Block_Size"])
        for r in keep:
            w.writerow([r.get("Dispatch_Id"), r.get("Kernel_Name", "")[:60], r.get("Start_Timestamp"), r.get("End_Timestamp"),
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")),
                        r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", "")])
print(json.dumps(res)[:1500])
