cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02pk
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
for cfg in ${CFGS}; do
  nz=${cfg%%:*}; g=${cfg##*:}
  extra=""
  if [ $nz = 150 ]; then extra="--ncol 50000"; fi
  MCKPP_KERNEL=pk MCKPP_PK=$g timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b_${nz}_$g.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r02pk/b_${nz}_$g.json"))
print("nz=$nz $g", "%.3e"%d['value'], "%.3f ms"%d['ms_per_step'], d['roofline']['kernel'])
PY
done
