# The round's records on the one-GPU box.  PART=a: smoke + the whole -m gpu suite (solver mode 0, then the suite
# again with MCKPP_SOLVER_MODE=1: library default and checker both) + the bench line; PART=b: rocprofv3 kernel trace
# and PMC counters, stamps, micro-benchmarks, geometry sweep.  Results under gpurun_out/r04final/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final
mkdir -p $O
if [ "${PART:-a}" = "a" ]; then
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
  MCKPP_SOLVER_MODE=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_solver_mode_1.log 2>&1; echo "pytest (MCKPP_SOLVER_MODE=1) rc=$?"; tail -3 $O/tests_solver_mode_1.log
  timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
  python tools/r04_digest.py $O/bench_n1.json | tee $O/bench_n1_digest.txt
  cp gpurun_out/parity_tolerance_gpu_*.json $O/ 2>/dev/null
else
  bash tools/r04_profile.sh 2>&1 | tail -12
  CFGS="40 60 69 100" bash tools/r04_stamp.sh 2>&1 | grep -v "wave-passes [0-9]\{4\};" | cut -c1-900 > $O/stamps.txt
  CFGS="69" BENCH_ARGS="--grid stretched --dto 1200 --land 0.35" bash tools/r04_stamp.sh 2>&1 | grep -v "wave-passes [0-9]\{4\};" | cut -c1-900 | sed -e 's/nz=69/nz=69 stretched grid, 35 % land, dto 1200/' >> $O/stamps.txt
  for u in sweeps lds issue lat; do [ -x tools/ubench/$u ] && timeout -k 5 120 tools/ubench/$u > $O/ubench_$u.txt 2>&1; done
  grep -E "two-ended|U,T,S forward|V forward" $O/ubench_sweeps.txt | grep "busy=0"
fi
