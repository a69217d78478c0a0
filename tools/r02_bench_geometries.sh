# CFGS="60:pk:8x2 69:ps:9x8x2 100:ps:auto 60:wg": one bench line per entry: <levels>:<kernel>[:<geometry>]
# geometry: MCKPP_PK=<waves>x<workgroups per CU> for pk, MCKPP_PS=<slots>x<waves>x<workgroups per CU> for ps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02pk
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
for cfg in ${CFGS}; do
  IFS=: read nz k g <<< "$cfg"
  extra=""
  if [ $nz -ge 150 ]; then extra="--ncol 50000"; fi
  if [ -z "$g" ] || [ "$g" = auto ]; then
    MCKPP_KERNEL=$k timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b.json 2>/dev/null
  elif [ $k = pk ]; then
    MCKPP_KERNEL=pk MCKPP_PK=$g timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b.json 2>/dev/null
  else
    MCKPP_KERNEL=ps MCKPP_PS=$g timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b.json 2>/dev/null
  fi
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02pk/b.json"))
    print("nz=$nz $k $g", "%.3e"%d['value'], "%.3f ms"%d['ms_per_step'], d['roofline']['kernel'])
except Exception as e:
    print("nz=$nz $k $g FAILED", e)
PY
done
