# Per-phase shader cycles of the manager wave: builds a profiling copy of the library with
# -DMCKPP_PK_STAMPS -DMCKPP_PS_STAMPS (cycle sums kept in registers of wave 0, flushed at kernel end) in /tmp,
# runs the bench with MCKPP_STAMP=1 on it, restores the product library.  k_column_wg stamps need no special build.
# CFGS entries: <levels>:<kernel>[:<geometry>] as in r02_bench_geometries.sh.
cd $GRAFT_REPO_ROOT
rm -rf /tmp/st && mkdir -p /tmp/st/mckpp_f90_amd /tmp/st/include
cp -r mckpp_f90_amd/csrc /tmp/st/mckpp_f90_amd/csrc && cp include/*.h /tmp/st/include/
rm -f /tmp/st/mckpp_f90_amd/csrc/*.o
make -C /tmp/st/mckpp_f90_amd/csrc EXTRA="-DMCKPP_PK_STAMPS -DMCKPP_PS_STAMPS" > /tmp/st/build.log 2>&1 || { tail -5 /tmp/st/build.log; exit 1; }
cp mckpp_f90_amd/libmckpp_hip.so /tmp/lib_keep.so
cp /tmp/st/mckpp_f90_amd/libmckpp_hip.so mckpp_f90_amd/libmckpp_hip.so
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2"
for cfg in ${CFGS:-60:pk 69:pk 100:pk 60:wg}; do
  IFS=: read nz k g <<< "$cfg"
  echo "== 1e5 columns x $nz levels, MCKPP_KERNEL=$k $g"
  if [ -z "$g" ] || [ "$g" = auto ]; then
    MCKPP_STAMP=1 MCKPP_KERNEL=$k timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "stamps" | tail -1
  elif [ $k = pk ]; then
    MCKPP_STAMP=1 MCKPP_KERNEL=pk MCKPP_PK=$g timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "stamps" | tail -1
  else
    MCKPP_STAMP=1 MCKPP_KERNEL=ps MCKPP_PS=$g timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "stamps" | tail -1
  fi
done
cp /tmp/lib_keep.so mckpp_f90_amd/libmckpp_hip.so
