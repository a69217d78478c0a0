# Forced geometries (MCKPP_PS=<slots>x<waves>x<workgroups per CU>) against the launcher's choice, per solver mode.
# CFGS entries <levels>:<geometry>[,<geometry>...]; MODES="0 1"; STEPS (default 20)
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-100:9x8x2}; do
  IFS=: read nz gs <<< "$cfg"
  for sm in ${MODES:-0 1}; do
    for g in chosen ${gs//,/ }; do
      if [ $g = chosen ]; then unset MCKPP_PS; else export MCKPP_PS=$g; fi
      echo "nz=$nz solver=$sm geometry $g"
      MCKPP_PS_VERBOSE=1 MCKPP_SOLVER_MODE=$sm timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-20} --warmup 3 --settle 0 --nz $nz 2>&1 | grep -E "mckpp ps|\"value\"" | sed -e 's/.*"value": \([0-9.e+]*\).*"ms_per_step": \([0-9.]*\).*/  rate \1 column-steps\/s, \2 ms per step/' | sort -u
    done
  done
done
unset MCKPP_PS
