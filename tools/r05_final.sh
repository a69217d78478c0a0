# The round's records on the one-GPU box.  PART=a: smoke + the -m gpu suite as it is and under its switches + the bench
# line (and profiles/config3_long_n1.json from it) + the N=2 rehearsal; PART=b: rocprofv3 kernel trace and PMC counters,
# stamps (full workgroups and a lone column), lone-pass times, micro-benchmarks.  Results under gpurun_out/r05final/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final
mkdir -p $O
if [ "${PART:-a}" = "a" ]; then
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests_gpu.log
  S="not 1000_steps and not full_length"
  MCKPP_SOLVER_MODE=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_solver_mode_1.log 2>&1; echo "pytest (MCKPP_SOLVER_MODE=1) rc=$?"; tail -2 $O/tests_solver_mode_1.log
  MCKPP_SOLO_AFTER=0 MCKPP_SOLO_LIMIT=1000000 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$S" > $O/tests_forced_solo.log 2>&1; echo "pytest (every lone column in the view of one slot) rc=$?"; tail -2 $O/tests_forced_solo.log
  MCKPP_SOLO_AFTER=0 MCKPP_SOLO_LIMIT=1000000 MCKPP_SOLVER_MODE=1 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "$S" > $O/tests_forced_solo_sm1.log 2>&1; echo "pytest (the same, solver mode 1) rc=$?"; tail -2 $O/tests_forced_solo_sm1.log
  MCKPP_SOLO=0 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$S" > $O/tests_solo_off.log 2>&1; echo "pytest (MCKPP_SOLO=0) rc=$?"; tail -2 $O/tests_solo_off.log
  MCKPP_MULTISTEP=0 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$S" > $O/tests_multistep0.log 2>&1; echo "pytest (MCKPP_MULTISTEP=0) rc=$?"; tail -2 $O/tests_multistep0.log
  MCKPP_PS_FIXED_L=0 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$S" > $O/tests_general_kernels.log 2>&1; echo "pytest (MCKPP_PS_FIXED_L=0) rc=$?"; tail -2 $O/tests_general_kernels.log
  MCKPP_L3_CAP=3 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$S" > $O/tests_l3cap3.log 2>&1; echo "pytest (MCKPP_L3_CAP=3) rc=$?"; tail -2 $O/tests_l3cap3.log
  timeout -k 10 800 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
  python tools/r05_digest.py $O/bench_n1.json | tee $O/bench_n1_digest.txt
  python3 - $O/bench_n1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ref = {"what": "config3_long of the N = 1 bench line (1e5 x 100, model steps 61-360 in one call): what config3_strong of an N > 1 line is compared with",
       "library_build": d["config"]["library_build"], "ms_per_step": d["config3_long"]["ms_per_step"], "value": d["config3_long"]["value"],
       "steps": d["config3_long"]["steps"], "workload": d["config3_long"]["workload"]}
json.dump(ref, open("gpurun_out/r05final/config3_long_n1.json", "w"), indent=1)
print("config3_long_n1:", ref["ms_per_step"], ref["library_build"])
PY
  cp gpurun_out/parity_tolerance_gpu_*.json $O/ 2>/dev/null
  OUT=r05final bash tools/r05_n2.sh
else
  bash tools/r05_profile.sh 2>&1 | tail -10
  CFGS="40 60 69 100" bash tools/r04_stamp.sh 2>&1 | grep -v "wave-passes [0-9]\{4\};" | cut -c1-900 > $O/stamps.txt
  CFGS="69" BENCH_ARGS="--grid stretched --dto 1200 --land 0.35" bash tools/r04_stamp.sh 2>&1 | grep -v "wave-passes [0-9]\{4\};" | cut -c1-900 | sed -e 's/nz=69/nz=69 stretched grid, 35 % land, dto 1200/' >> $O/stamps.txt
  bash tools/r05_lone_stamps.sh 2>&1 | grep -A1 "wave-passes 20[0-9];" | grep -v "^--" | cut -c1-900 > $O/lone_stamps.txt
  LONE_REPS=3 timeout -k 10 500 python tools/r05_lone_probe.py 60 100 > $O/lone.txt 2>&1
  MCKPP_SOLO=0 LONE_REPS=3 timeout -k 10 500 python tools/r05_lone_probe.py 60 100 > $O/lone_solo_off.txt 2>&1
  for u in sweeps lds issue lat; do [ -x tools/ubench/$u ] && timeout -k 5 120 tools/ubench/$u > $O/ubench_$u.txt 2>&1; done
  grep -E "two-ended|U,T,S forward|V forward" $O/ubench_sweeps.txt | grep "busy=0"
fi
