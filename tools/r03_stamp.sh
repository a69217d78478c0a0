# Per-phase shader cycles of the manager wave: runs the bench on stamped builds of the library
# (-DMCKPP_PS_STAMPS, made by tools/r03_build_variants.sh as .ab/lib<name>.so) through MCKPP_HIP_LIBRARY.
# LIBS="S" (default), CFGS entries: <levels>[:<geometry>] as in r02_geometry.sh, BENCH_ARGS as in r03_ab.sh.
cd $GRAFT_REPO_ROOT
for v in ${LIBS:-S}; do
  B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 $BENCH_ARGS"
  for cfg in ${CFGS:-40 60 69 100}; do
    IFS=: read nz g <<< "$cfg"
    echo "== $v nz=$nz $g"
    if [ -z "$g" ]; then
      MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so MCKPP_PS_VERBOSE=1 MCKPP_STAMP=1 timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "mckpp ps|stamps" | tail -2
    else
      MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so MCKPP_PS=$g MCKPP_PS_VERBOSE=1 MCKPP_STAMP=1 timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "mckpp ps|stamps" | tail -2
    fi
  done
done
