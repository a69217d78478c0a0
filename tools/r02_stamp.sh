cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
for cfg in "60 4x4" "60 8x2" "69 8x2" "100 8x2" "100 4x4"; do
  set -- $cfg
  echo "== nz=$1 pk=$2"
  MCKPP_STAMP=1 MCKPP_KERNEL=pk MCKPP_PK=$2 timeout -k 10 200 $B --nz $1 2>&1 | grep "stamps" | tail -1
done
echo "== wg nz=60"; MCKPP_STAMP=1 timeout -k 10 200 $B --nz 60 2>&1 | grep "stamps" | tail -1
