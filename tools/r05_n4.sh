# Rehearsal of the N=4 line on the one GPU (every rank on device 0, gloo), as the driver would start it, timed; and the
# wall time of the default N=1 line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r05k}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
MCKPP_BENCH_SHARE_GPU=1 MCKPP_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 20 --warmup 3 > $O/bench_n4_shared.json 2> $O/bench_n4_shared.err; echo "N=4 rc=$? wall $(( $(date +%s) - t0 )) s"
python3 - $O/bench_n4_shared.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); m = d["multi_gpu"]
print("N=4 shared: value %.4g, ms/step %.3f, ok=%s, per-rank %s" % (d["value"], d["ms_per_step"], d.get("ok"), [round(x, 2) for x in m["per_rank_ms_per_step"]["all"]]))
print("  gather:", {k: v for k, v in m["gather"].items() if k in ("error", "checked", "T_ms", "hmix_ms")})
print("  config3_strong:", {k: v for k, v in m["config3_strong"].items() if k in ("error", "ms_per_step", "per_rank_ms_per_step", "value", "ranks")})
print("  single_process:", str(m.get("single_process"))[:200])
print("  cpu_baseline:", d.get("cpu_baseline", {}).get("value"), "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
PY
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1_driver_args.json 2> $O/bench_n1_driver_args.err; echo "N=1 (--steps 20 --warmup 3) rc=$? wall $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d=json.load(open('$O/bench_n1_driver_args.json')); r=d['roofline']; print('N=1: value %.4g ms %.3f frac %.4f traffic %s frac_wo_diag %.4f cpu %.4g' % (d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['frac_without_diagnostic_bytes'], d['cpu_baseline']['value']))"
