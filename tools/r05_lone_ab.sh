cd $GRAFT_REPO_ROOT
for v in B N; do
  echo "== lib $v"
  MCKPP_HIP_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/r05_lone_probe.py 60 100 2>&1 | grep "solver=0" | grep "launcher\|15x8x2\|19x16x1"
done
