# Per-phase shader cycles of the manager wave: builds a profiling copy of the library with -DMCKPP_PS_STAMPS
# (cycle sums kept in registers of wave 0, flushed at kernel end) in /tmp, runs the bench with MCKPP_STAMP=1 on
# it, restores the product library.  CFGS entries: <levels>[:<geometry>] as in r02_geometry.sh.
cd $GRAFT_REPO_ROOT
rm -rf /tmp/st && mkdir -p /tmp/st/mckpp_f90_amd /tmp/st/include
cp -r mckpp_f90_amd/csrc /tmp/st/mckpp_f90_amd/csrc && cp include/*.h /tmp/st/include/
rm -f /tmp/st/mckpp_f90_amd/csrc/*.o
make -C /tmp/st/mckpp_f90_amd/csrc EXTRA="-DMCKPP_PS_STAMPS" > /tmp/st/build.log 2>&1 || { tail -5 /tmp/st/build.log; exit 1; }
cp mckpp_f90_amd/libmckpp_hip.so /tmp/lib_keep.so
cp /tmp/st/mckpp_f90_amd/libmckpp_hip.so mckpp_f90_amd/libmckpp_hip.so
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 $BENCH_ARGS"   # BENCH_ARGS: e.g. "--grid stretched --dto 1200 --land 0.35"
for cfg in ${CFGS:-40 60 69 100}; do
  IFS=: read nz g <<< "$cfg"
  if [ -z "$g" ]; then
    MCKPP_PS_VERBOSE=1 MCKPP_STAMP=1 timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "mckpp ps|stamps" | tail -2
  else
    MCKPP_PS=$g MCKPP_PS_VERBOSE=1 MCKPP_STAMP=1 timeout -k 10 200 $B --nz $nz 2>&1 | grep -E "mckpp ps|stamps" | tail -2
  fi
done
cp /tmp/lib_keep.so mckpp_f90_amd/libmckpp_hip.so
