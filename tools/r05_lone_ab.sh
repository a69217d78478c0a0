# lone-pass A/B of library builds (.ab/lib<name>.so): LIBS="B N", best of LONE_REPS runs per configuration
cd $GRAFT_REPO_ROOT
for v in ${LIBS:-B N}; do
  echo "== lib $v"
  MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so LONE_REPS=${LONE_REPS:-4} timeout -k 10 500 python tools/r05_lone_probe.py ${NZS:-60 100} 2>&1 | grep "solver=${SM:-0}" | grep "launcher\|15x8x2\|19x16x1"
done
