# Is the rate a function of how long the run is (sustained clock)?  CFGS as in r02_bench_geometries.sh; STEPS list.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02pk
for cfg in ${CFGS}; do
  IFS=: read nz k g <<< "$cfg"
  for st in ${STEPS:-5 20 60}; do
    B="python bench.py --no-cpu-baseline --no-extras --steps $st --warmup 2 --nz $nz"
    ( sleep 6; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ';' > gpurun_out/r02pk/smi.txt ) &
    if [ $k = pk ]; then MCKPP_KERNEL=pk MCKPP_PK=$g timeout -k 10 200 $B > gpurun_out/r02pk/b.json 2>/dev/null
    else MCKPP_KERNEL=ps MCKPP_PS=$g timeout -k 10 200 $B > gpurun_out/r02pk/b.json 2>/dev/null; fi
    wait
    python - <<PY
import json
d=json.load(open("gpurun_out/r02pk/b.json"))
print("nz=$nz $k $g steps=$st", "%.3e"%d['value'], "%.3f ms"%d['ms_per_step'], "kernel avg %.3f ms" % d['roofline'].get('kernel_avg_ms', -1), open("gpurun_out/r02pk/smi.txt").read()[:300])
PY
  done
done
