# The driver's bench runs, on the one-GPU box: `python bench.py` (full line incl. cpu_baseline) and the N=2
# rehearsal started plainly (no launcher; both ranks on device 0, collectives on gloo).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03bench
( time python bench.py ) > gpurun_out/r03bench/bench_n1.json 2> gpurun_out/r03bench/bench_n1.err
tail -4 gpurun_out/r03bench/bench_n1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03bench/bench_n1.json").read().strip().splitlines()[-1])
print("value %.3e ms %.3f frac %.4f kernel %s build %s"%(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel'],d['config']['library_build']))
print("sustained", d['sustained'])
print("drop_in", {k:(v if isinstance(v,str) else {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()}) for k,v in d['drop_in'].items() if k!='what'})
print("diurnal", {k:v for k,v in d['diurnal'].items() if k not in('workload','cpu_port')})
for s in d['other_shapes']: print({k:v for k,v in s.items()})
print(d['strong_scaling_proxy'])
print(d['cpu_baseline'])
PY
if [ -z "$SKIP_N2" ]; then
( time MCKPP_BENCH_SHARE_GPU=1 MCKPP_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 2 --ncol 50000 ) > gpurun_out/r03bench/bench_n2_shared.json 2> gpurun_out/r03bench/bench_n2_shared.err
echo "N=2 rc=$?"; tail -3 gpurun_out/r03bench/bench_n2_shared.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03bench/bench_n2_shared.json").read().strip().splitlines()[-1])
print(d['n_gpus'], '%.3e'%d['value'], d['ms_per_step'], d['scaling'])
print(json.dumps(d['multi_gpu'], indent=1)[:2500])
PY
( MCKPP_BENCH_SHARE_GPU=1 MCKPP_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 2 --total-ncol 100000 --nz 100 ) > gpurun_out/r03bench/bench_n2_strong.json 2> gpurun_out/r03bench/bench_n2_strong.err
echo "N=2 strong rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03bench/bench_n2_strong.json").read().strip().splitlines()[-1])
print(d['n_gpus'], '%.3e'%d['value'], d['ms_per_step'], d['scaling'], d['config']['workload'][:80])
PY
fi
