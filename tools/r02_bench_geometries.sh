cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02pk
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
if [ -n "$PKTEST" ]; then timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "variant or tiny or trap or zero_pivot or uniform_grid or seeded or long_iter" 2>&1 | tail -1; fi
for cfg in ${CFGS}; do
  nz=${cfg%%:*}; g=${cfg##*:}
  extra=""
  if [ $nz = 150 ]; then extra="--ncol 50000"; fi
  if [ $g = wg ]; then
    timeout -k 10 200 $B --nz $nz > gpurun_out/r02pk/b_${nz}_$g.json 2>/dev/null
  else
    MCKPP_KERNEL=pk MCKPP_PK=$g timeout -k 10 200 $B --nz $nz $extra > gpurun_out/r02pk/b_${nz}_$g.json 2>/dev/null
  fi
  python - <<PY
import json
d=json.load(open("gpurun_out/r02pk/b_${nz}_$g.json"))
print("nz=$nz $g", "%.3e"%d['value'], "%.3f ms"%d['ms_per_step'], d['roofline']['kernel'])
PY
done
