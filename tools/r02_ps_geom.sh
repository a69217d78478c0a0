# prints the geometry k_column_ps runs with and its rate: CFGS="<levels>[:<geometry>] ..."
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS}; do
  IFS=: read nz g <<< "$cfg"
  if [ -z "$g" ]; then MCKPP_PS_VERBOSE=1 MCKPP_KERNEL=ps timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --nz $nz 2>&1 | grep -E "mckpp ps|\"value\"" | sed -e 's/.*"value": \([0-9.e+]*\).*/rate \1/'
  else MCKPP_PS=$g MCKPP_PS_VERBOSE=1 MCKPP_KERNEL=ps timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --nz $nz 2>&1 | grep -E "mckpp ps|\"value\"" | sed -e 's/.*"value": \([0-9.e+]*\).*/rate \1/'; fi
done
