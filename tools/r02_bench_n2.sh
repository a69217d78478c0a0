# rehearsal of the N=2 launch on a one-GPU box (both ranks on device 0, collectives on gloo)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bench
MCKPP_BENCH_BACKEND=gloo MCKPP_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --ncol 50000 > gpurun_out/r02bench/bench_n2_shared.json 2> gpurun_out/r02bench/bench_n2_shared.err
tail -2 gpurun_out/r02bench/bench_n2_shared.err
python -c "
import json
d=json.loads(open('gpurun_out/r02bench/bench_n2_shared.json').read().strip().splitlines()[-1])
print(d['n_gpus'], '%.3e'%d['value'], d['ms_per_step'], d['config']['sharding'], list(d.keys()))"
