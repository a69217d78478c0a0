# Rehearsal of the N>1 line on the one GPU (both ranks on device 0, gloo): as it is, and with a failure on rank 1 alone
# in the diagnostics download / in the config3_strong block
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-r05g}; mkdir -p $O
export MCKPP_BENCH_SHARE_GPU=1 MCKPP_BENCH_BACKEND=gloo
A="--gpus 2 --steps 10 --warmup 2 --settle 40 --ncol 50000"
timeout -k 10 500 python3 bench.py $A > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "rc=$?"
MCKPP_BENCH_FAIL_RANK=1 MCKPP_BENCH_FAIL_RANK3=1 timeout -k 10 500 python3 bench.py $A --no-cpu-baseline > $O/bench_n2_fail.json 2> $O/bench_n2_fail.err; echo "rc(fail on rank 1)=$?"
python3 - $O <<'PY'
import json, sys
for f in ("bench_n2_shared.json", "bench_n2_fail.json"):
    d = json.load(open(sys.argv[1] + "/" + f)); m = d["multi_gpu"]
    print(f, "value %.4g ok=%s" % (d["value"], d.get("ok")), "| gather:", {k: v for k, v in m["gather"].items() if k in ("error", "checked", "T_ms")},
          "| config3_strong:", {k: v for k, v in m["config3_strong"].items() if k in ("error", "ms_per_step", "per_rank_ms_per_step", "value", "n1_reference")})
PY
