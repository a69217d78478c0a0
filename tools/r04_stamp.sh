# Per-phase shader cycles of the manager wave (a -DMCKPP_PS_STAMPS build .ab/lib<name>.so), per solver mode.
# LIBS="S", MODES="0 1", CFGS entries <levels>[:<geometry>], BENCH_ARGS.
cd $GRAFT_REPO_ROOT
for v in ${LIBS:-S}; do
  B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --settle 0 $BENCH_ARGS"
  for sm in ${MODES:-0 1}; do
    for cfg in ${CFGS:-60}; do
      IFS=: read nz g <<< "$cfg"
      echo "== $v solver=$sm nz=$nz $g"
      export MCKPP_SOLVER_MODE=$sm MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so MCKPP_PS_VERBOSE=1 MCKPP_STAMP=1
      if [ -n "$g" ]; then export MCKPP_PS=$g; else unset MCKPP_PS; fi
      timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "mckpp ps|stamps" | tail -2
    done
  done
done
unset MCKPP_SOLVER_MODE MCKPP_HIP_LIBRARY MCKPP_PS_VERBOSE MCKPP_STAMP MCKPP_PS
