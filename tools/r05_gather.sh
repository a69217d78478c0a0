# Round 5, lagging columns gathered: the long legs under several values of MCKPP_GATHER
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r05v}; mkdir -p $O
for g in ${GS:-0 1 2 4 8}; do
  MCKPP_GATHER=$g timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle 0 --legs config3_long_12500${LEGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('config3_long_12500','config3_long','config3_long_two_ended_solver'):
    if k in d and 'ms_per_step' in d[k]:
        c=d[k]['census']; print('gather $g %s: %.3f ms/step multi; single-step launches %.2f ms' % (k, d[k]['ms_per_step'], c['ms_per_step_mean']))
print('gather $g headline %.3f ms' % d['ms_per_step'])"
done
