cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2"
for cfg in ${CFGS:-"60 8x2" "100 7x2"}; do
  set -- $cfg
  echo "== nz=$1 pk=$2"
  MCKPP_STAMP=1 MCKPP_KERNEL=pk MCKPP_PK=$2 timeout -k 10 200 $B --nz $1 2>&1 | grep -E "stamps" | tail -1
  MCKPP_STAMP=1 MCKPP_KERNEL=pk MCKPP_PK=$2 timeout -k 10 200 $B --nz $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step %.3f'%d['ms_per_step'])"
done
