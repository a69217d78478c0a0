"""One screen of a bench.py line: python tools/r05_digest.py bench.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"value {d['value']:.4g} {d['unit']}  {d['ms_per_step']:.3f} ms/step  ({d['n_gpus']} GPU, {d['steps']} steps, solver: {d['config'].get('solver', '?')[:40]})")
print(f"  roofline frac {r['frac']:.4f} = {r['achieved']:.0f} GB/s of {r['peak']:.0f}; kernel {r['kernel_avg_ms']:.3f} ms; traffic {r.get('traffic')}")
if "burst" in d:
    print(f"  burst (cold device): {d['burst']['value_this_rank']:.4g}, {d['burst']['ms_per_step']:.3f} ms/step")
print(f"  column passes/s {d.get('column_passes_per_s', 0):.4g}; passes mean {d['config']['mean_passes_per_column_step_last_step']:.3f} max {d['config']['max_passes_last_step']}")
for k in ("two_ended_solver", "sustained"):
    if k in d:
        print(f"  {k}: {d[k]['value']:.4g}, {d[k]['ms_per_step']:.3f} ms/step" + (f", x{d[k]['config']['ratio_to_value']:.3f} of value" if k == "two_ended_solver" else ""))
if "tail" in d:
    t = d["tail"]
    print(f"  tail: a launch per step {t['value']:.4g} ({t['ms_per_step']:.3f} ms/step; {t['steps_with_a_column_over_50_passes']} of {t['steps']} steps "
          f"have a column over 50 passes: {t['ms_of_those_steps']} ms, the others {t['ms_of_the_other_steps_mean']:.3f}); the same steps as one launch "
          f"{t['the_same_steps_as_one_launch']['value']:.4g} ({t['the_same_steps_as_one_launch']['ms_per_step']:.3f} ms/step)")
if "config1_pass" in d:
    c = d["config1_pass"]
    print(f"  configs[1]: {c['value']:.4g} column-passes/s, kernel {c['kernel_avg_ms']:.4f} ms, roofline frac {c['roofline']['frac']:.4f}")
if "drop_in" in d:
    di = d["drop_in"]
    print("  drop-in ms/step: " + ", ".join(f"{k} {v['ms_per_step']:.2f}" for k, v in di.items() if isinstance(v, dict)))
for o in d.get("other_shapes", []):
    print(f"  {o['workload'][:70]}: {o['value']:.4g}, frac {o['roofline_frac']:.4f}, passes {o['mean_passes_per_column_step_last_step']:.2f}")
for k in ("config3_long", "config3_long_12500", "config3_long_two_ended_solver"):
    if k in d:
        c = d[k]["census"]
        print(f"  {k}: {d[k]['ms_per_step']:.3f} ms/step in one call ({d[k]['value']:.4g}); the steps after it one by one: {c['ms_per_step_mean']:.2f} ms "
              f"({c['ms_per_step_min']:.2f}-{c['ms_per_step_max']:.2f}); columns > 50 passes per step {c['columns_over_50_passes_per_step']['mean']:.0f}")
if "strong_scaling_proxy" in d:
    sp = d["strong_scaling_proxy"]
    print(f"  strong-scaling proxy: nz100 {sp['nz100']['ratio_to_1e5']:.3f}, nz60 {sp['nz60']['ratio_to_1e5']:.3f}")
if "diurnal" in d:
    print(f"  diurnal: {d['diurnal']['value']:.4g}; cpu port {d['diurnal'].get('cpu_port', {}).get('value', 0):.4g}")
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"  cpu_baseline: {c['value']:.4g} on {c['cores']} cores ({c['kind']}); 1 thread {c.get('value_1thread', 0):.4g}")
