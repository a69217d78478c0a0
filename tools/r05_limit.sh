cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
for lim in 8 32 128 100000; do
  MCKPP_SOLO_LIMIT=$lim timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle 0 --legs config3_long_12500 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['config3_long_12500']; c=k['census']
print('limit $lim 12500: %.3f ms/step multi; single-step launches %.2f ms' % (k['ms_per_step'], c['ms_per_step_mean']))"
done
for lim in 32 128; do
  MCKPP_SOLO_LIMIT=$lim timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle 0 --legs config3_long 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['config3_long']; c=k['census']
print('limit $lim 1e5: %.3f ms/step multi; single-step launches %.2f ms' % (k['ms_per_step'], c['ms_per_step_mean']))"
done
