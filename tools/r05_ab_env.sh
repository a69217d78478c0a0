# A/B of an environment switch on one box: VAR=<name> VALS="a b", headline rate at CFGS levels (+ long legs with LONG=1)
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${ROUNDS:-2}); do for v in $VALS; do for nz in ${CFGS:-60}; do
  env $VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-20} --warmup 3 --settle ${SETTLE:-200} --nz $nz 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v nz=$nz value %.4g ms %.3f burst %.3f maxpass %d' % (d['value'], d['ms_per_step'], d['burst']['ms_per_step'], d['config']['max_passes_last_step']))"
done; done; done
if [ "${LONG:-0}" = 1 ]; then for v in $VALS; do
  env $VAR=$v timeout -k 10 500 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle 0 --legs config3_long,config3_long_12500 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('config3_long','config3_long_12500'):
    c=d[k]['census']; print('$VAR=$v %s: %.3f ms/step multi; single-step launches %.2f ms (min %.2f max %.2f)' % (k, d[k]['ms_per_step'], c['ms_per_step_mean'], c['ms_per_step_min'], c['ms_per_step_max']))"
done; fi
