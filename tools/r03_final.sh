# Everything the round's records need, on the one-GPU box (PART=a: tests + bench lines; PART=b: profiles, stamps,
# rates of the kernel variants, micro-benchmarks).  Results under gpurun_out/r03final/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final
mkdir -p $O
if [ "${PART:-a}" = "a" ]; then
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/tests.log
  bash tools/r03_bench.sh 2>&1 | tail -40
  cp gpurun_out/r03bench/*.json $O/ 2>/dev/null
else
  bash tools/r03_profile.sh 2>&1 | grep -E "done|du|prof_r03" | tail -6
  LIBS=S CFGS="40 60 69 100" bash tools/r03_stamp.sh 2>&1 | cut -c1-420 > $O/stamps.txt
  LIBS=S CFGS="69" BENCH_ARGS="--grid stretched --dto 1200 --land 0.35" bash tools/r03_stamp.sh 2>&1 | cut -c1-420 | sed -e 's/nz=69/nz=69 stretched grid, 35 % land, dto 1200/' >> $O/stamps.txt
  tail -4 $O/stamps.txt | cut -c1-200
  for nz in 60 100; do python tools/r03_ext_rate.py $nz 2>/dev/null; done | tee $O/variants.txt
  CFGS="20 40 60 69 80 100 125 150" STEPS=10 bash tools/r03_geometry.sh > $O/geometry.txt 2>&1; tail -2 $O/geometry.txt
  for u in sweeps lds issue lat; do [ -x tools/ubench/$u ] && timeout -k 5 120 tools/ubench/$u > $O/ubench_$u.txt 2>&1; done
  head -3 $O/ubench_sweeps.txt
fi
