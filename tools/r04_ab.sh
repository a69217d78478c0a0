# A/B of library builds and solver modes on one box: .ab/lib<name>.so (tools/r03_build_variants.sh) through
# MCKPP_HIP_LIBRARY.  LIBS="A B", MODES="0 1" (MCKPP_SOLVER_MODE), CFGS="<levels> ...", ROUNDS (default 2),
# STEPS (default 20), BENCH_ARGS.
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${LIBS:-A}; do
    for sm in ${MODES:-0 1}; do
      for nz in ${CFGS:-60}; do
        MCKPP_SOLVER_MODE=$sm MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-20} --warmup 3 --nz $nz $BENCH_ARGS 2>/dev/null | sed -e "s/.*\"value\": \([0-9.e+]*\).*\"ms_per_step\": \([0-9.]*\).*/$v solver=$sm nz=$nz rate \1 column-steps\/s, \2 ms per step/"
      done
    done
  done
done
