# Prints the geometry k_column_ps runs with and the rate it gives: CFGS="<levels>[:<geometry>] ..."
# geometry = MCKPP_PS=<slots>x<waves>x<workgroups per CU> (overrides the launcher's choice); STEPS (default 10).
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS}; do
  IFS=: read nz g <<< "$cfg"
  extra=""
  if [ $nz -ge 150 ]; then extra="--ncol 50000"; fi
  B="python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-10} --warmup 2 --nz $nz $extra"
  if [ -z "$g" ]; then MCKPP_PS_VERBOSE=1 timeout -k 10 200 $B 2>&1 | grep -E "mckpp ps|\"value\"" | sed -e 's/.*"value": \([0-9.e+]*\).*"ms_per_step": \([0-9.]*\).*/rate \1 column-steps\/s, \2 ms per step/'
  else MCKPP_PS=$g MCKPP_PS_VERBOSE=1 timeout -k 10 200 $B 2>&1 | grep -E "mckpp ps|\"value\"" | sed -e 's/.*"value": \([0-9.e+]*\).*"ms_per_step": \([0-9.]*\).*/rate \1 column-steps\/s, \2 ms per step/'; fi
done
