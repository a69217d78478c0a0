# The round's bench lines: python bench.py (full line, N=1), its digest, and the rehearsal of `bench.py --gpus 2`
# started plainly on the one GPU (both ranks on device 0, gloo): weak and strong mode.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04bench
mkdir -p $O
timeout -k 10 700 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python tools/r04_digest.py $O/bench_n1.json | tee $O/bench_n1_digest.txt
MCKPP_BENCH_SHARE_GPU=1 MCKPP_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 --ncol 50000 > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "bench n2 rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_n2_shared.json')); print('N=2 (shared GPU, gloo): value %.4g, n_gpus %d, roofline frac %.4f, cpu_baseline %.4g on %d cores (%s), per-rank ms %s, single_process %s' % (d['value'], d['n_gpus'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:50], d['multi_gpu']['per_rank_ms_per_step']['all'], json.dumps(d['multi_gpu']['single_process'])[:300]))"
MCKPP_BENCH_SHARE_GPU=1 MCKPP_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 --nz 100 --total-ncol 100000 --no-cpu-baseline > $O/bench_n2_strong.json 2> $O/bench_n2_strong.err; echo "bench n2 strong rc=$?"
