# config3_long legs of bench.py under MCKPP_SOLO=0 (the round-4 behaviour) and the default
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r05d}; mkdir -p $O
for solo in ${SOLOS:-0 1}; do
  MCKPP_SOLO=$solo timeout -k 10 500 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle 0 --legs ${LEGS:-config3_long,config3_long_12500} > $O/long_solo$solo.json 2>$O/long_solo$solo.err
  python3 - $O/long_solo$solo.json $solo <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("config3_long", "config3_long_12500", "config3_long_two_ended_solver"):
    if k in d:
        c = d[k]["census"]
        print(f"solo={sys.argv[2]} {k}: {d[k]['ms_per_step']:.3f} ms per step in one call ({d[k]['value']:.4g} column-steps/s); census: {c['ms_per_step_mean']:.2f} ms per single-step launch, "
              f"columns > 50 passes per step {c['columns_over_50_passes_per_step']}, share of steps with an itermax column {c['share_of_steps_with_a_column_at_itermax']:.2f}")
PY
done
