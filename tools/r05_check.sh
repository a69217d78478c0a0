# Round 5: the -m gpu suite as it is and with the view of one slot forced on every column that is left alone in its
# workgroup (MCKPP_SOLO_AFTER=0 MCKPP_SOLO_LIMIT=1000000: every workgroup drains after every round), then rates.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r05b}; mkdir -p $O
if [ "${TESTS:-1}" = 1 ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "suite rc=$?"; tail -3 $O/tests_gpu.log
  MCKPP_SOLO_AFTER=0 MCKPP_SOLO_LIMIT=1000000 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not 1000_steps and not full_length" > $O/tests_forced_solo.log 2>&1; echo "forced solo rc=$?"; tail -3 $O/tests_forced_solo.log
  MCKPP_SOLO=0 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "not 1000_steps and not full_length" > $O/tests_solo_off.log 2>&1; echo "solo off rc=$?"; tail -3 $O/tests_solo_off.log
fi
if [ "${RATES:-1}" = 1 ]; then
  timeout -k 10 400 python tools/r05_lone_probe.py 60 100 > $O/lone.txt 2>&1; grep "launcher\|15x8x2\|19x16x1" $O/lone.txt
  for nz in 60 100; do for solo in 0 1; do
    MCKPP_SOLO=$solo timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --settle ${SETTLE:-200} --nz $nz 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nz=$nz solo=$solo value %.4g ms %.3f burst %.3f maxpass %d' % (d['value'], d['ms_per_step'], d['burst']['ms_per_step'], d['config']['max_passes_last_step']))"
  done; done
fi
