# per-phase shader cycles of the manager wave (k_column_pk built with -DMCKPP_PK_STAMPS) and of wave 1 (k_column_wg)
cd $GRAFT_REPO_ROOT
cp mckpp_f90_amd/libmckpp_hip.so /tmp/lib_keep.so
cp tools/ab/lib_stamps.so mckpp_f90_amd/libmckpp_hip.so
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2"
for cfg in "60 pk" "69 pk" "100 pk" "60 wg"; do
  set -- $cfg
  echo "== 1e5 columns x $1 levels, MCKPP_KERNEL=$2"
  MCKPP_STAMP=1 MCKPP_KERNEL=$2 timeout -k 10 200 $B --nz $1 2>&1 | grep -E "stamps" | tail -1
done
cp /tmp/lib_keep.so mckpp_f90_amd/libmckpp_hip.so
