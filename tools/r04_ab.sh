# A/B of library builds and solver modes on one box: .ab/lib<name>.so (tools/r03_build_variants.sh) through
# MCKPP_HIP_LIBRARY.  LIBS="A B", MODES="0 1" (MCKPP_SOLVER_MODE), CFGS="<levels> ...", ROUNDS (default 2),
# STEPS (default 20), SETTLE (default 0: the cold device's burst, as the earlier rounds' A/B runs), BENCH_ARGS.
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${LIBS:-A}; do
    for sm in ${MODES:-0 1}; do
      for nz in ${CFGS:-60}; do
        MCKPP_SOLVER_MODE=$sm MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-20} --warmup 3 --settle ${SETTLE:-0} --nz $nz $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); b=d.get('burst'); print('$v solver=$sm nz=$nz rate %.4g column-steps/s, %.3f ms per step, passes %.2f max %d' % (d['value'], d['ms_per_step'], d['config']['mean_passes_per_column_step_last_step'], d['config']['max_passes_last_step']) + (' (cold burst before the settle leg: %.3f ms)' % b['ms_per_step'] if b else ''))"
      done
    done
  done
done
