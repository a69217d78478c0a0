set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02base
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
$B > gpurun_out/r02base/nz60_wg.json
MCKPP_KERNEL=mw $B > gpurun_out/r02base/nz60_mw1.json
$B --nz 69 > gpurun_out/r02base/nz69.json
$B --nz 100 > gpurun_out/r02base/nz100.json
MCKPP_MW=4x2x4 $B --nz 69 > gpurun_out/r02base/nz69_w4.json
MCKPP_MW=4x2x4 $B --nz 100 > gpurun_out/r02base/nz100_w4.json
$B --ncol 12500 --nz 100 > gpurun_out/r02base/nz100_12k.json
$B --ncol 12500 --nz 60 > gpurun_out/r02base/nz60_12k.json
$B --nz 40 > gpurun_out/r02base/nz40.json
for f in gpurun_out/r02base/*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['config']['mean_passes_per_column_step_last_step'])"; done
