# CFGS="60:8x2 69:8x2 100:7x2 60:wg 100:auto": one bench line per entry; <levels>:<MCKPP_PK geometry | wg | auto>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02pk
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
for cfg in ${CFGS}; do
  nz=${cfg%%:*}; g=${cfg##*:}
  extra=""
  if [ $nz -ge 150 ]; then extra="--ncol 50000"; fi
  if [ $g = wg ]; then
    MCKPP_KERNEL=wg timeout -k 10 200 $B --nz $nz > gpurun_out/r02pk/b_${nz}_$g.json 2>/dev/null
  elif [ $g = auto ]; then
    MCKPP_KERNEL=pk timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b_${nz}_$g.json 2>/dev/null
  else
    MCKPP_KERNEL=pk MCKPP_PK=$g timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b_${nz}_$g.json 2>/dev/null
  fi
  python - <<PY
import json
d=json.load(open("gpurun_out/r02pk/b_${nz}_$g.json"))
print("nz=$nz $g", "%.3e"%d['value'], "%.3f ms"%d['ms_per_step'], d['roofline']['kernel'])
PY
done
