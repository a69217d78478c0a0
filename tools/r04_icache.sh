# Instruction-cache counters of the column kernel (its loop body is ~85 KB of code): one --pmc pass on a short bench run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/icache_r04
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/names.txt
echo "available: $(cat $OUT/names.txt)"
for sm in ${MODES:-0}; do
  export MCKPP_SOLVER_MODE=$sm
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/ic_sm$sm -- python3 bench.py --steps 3 --warmup 2 --settle 0 --no-cpu-baseline --no-extras ${BENCH_ARGS} > $OUT/ic_sm$sm.json 2> $OUT/ic_sm$sm.err
  python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); n=0
for f in glob.glob("$OUT/ic_sm$sm/**/*counter_collection.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    # the last k_column_ps dispatch (the timed 3 steps)
    ids=[r["Dispatch_Id"] for r in rows if "k_column_ps" in r["Kernel_Name"]]
    last=ids[-1] if ids else None
    for r in rows:
        if r["Dispatch_Id"]==last: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print("solver mode $sm, the timed dispatch (3 steps):", dict(tot))
if tot.get("SQC_ICACHE_REQ"): print("  hit rate %.4f, misses per step %.3g" % (tot["SQC_ICACHE_HITS"]/tot["SQC_ICACHE_REQ"], tot.get("SQC_ICACHE_MISSES",0)/3))
PY
done
