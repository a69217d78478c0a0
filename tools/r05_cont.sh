# Round 5, continuation of lagging columns: tests (suite as it is and with views forced), then the long legs under several
# values of the drain limit.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r05v}; mkdir -p $O
if [ "${TESTS:-1}" = 1 ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/tests_gpu.log
  [ $rc = 0 ] || exit 1
  MCKPP_SOLO_AFTER=0 MCKPP_SOLO_LIMIT=1000000 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not 1000_steps and not full_length" > $O/tests_forced_solo.log 2>&1; rc=$?; echo "forced solo rc=$rc"; tail -3 $O/tests_forced_solo.log
  [ $rc = 0 ] || exit 1
fi
for lim in ${LIMS:-8 32 128 100000}; do
  MCKPP_SOLO_LIMIT=$lim timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle 0 --legs config3_long_12500${LEGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('config3_long_12500','config3_long'):
    if k in d and 'ms_per_step' in d[k]:
        c=d[k]['census']; print('limit $lim %s: %.3f ms/step multi; single-step launches %.2f ms' % (k, d[k]['ms_per_step'], c['ms_per_step_mean']))
print('limit $lim headline %.3f ms' % d['ms_per_step'])"
done
