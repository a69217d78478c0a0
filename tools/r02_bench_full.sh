cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bench
( time python bench.py ) > gpurun_out/r02bench/bench_n1.json 2> gpurun_out/r02bench/bench_n1.err
tail -5 gpurun_out/r02bench/bench_n1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02bench/bench_n1.json").read().strip().splitlines()[-1])
print("value %.3e ms %.3f frac %.4f kernel %s"%(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel']))
print("diurnal", {k:v for k,v in d['diurnal'].items() if k not in('workload',)})
for s in d['other_shapes']: print(s)
print(d['strong_scaling_proxy'])
print(d['cpu_baseline'])
PY
