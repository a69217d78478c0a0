# A/B of two builds of the library on one box: .ab/libA.so and .ab/libB.so (made by hand from two source
# states) take turns as the product library; CFGS="<levels> ..." as in r02_geometry.sh, ROUNDS repeats (default 2).
cd $GRAFT_REPO_ROOT
cp mckpp_f90_amd/libmckpp_hip.so /tmp/lib_keep.so
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${LIBS:-A B}; do
    cp .ab/lib$v.so mckpp_f90_amd/libmckpp_hip.so
    for nz in ${CFGS:-60}; do
      python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-20} --warmup 3 --nz $nz $BENCH_ARGS 2>/dev/null | sed -e "s/.*\"value\": \([0-9.e+]*\).*\"ms_per_step\": \([0-9.]*\).*/$v nz=$nz rate \1 column-steps\/s, \2 ms per step/"
    done
  done
done
cp /tmp/lib_keep.so mckpp_f90_amd/libmckpp_hip.so
