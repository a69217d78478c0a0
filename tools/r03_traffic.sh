# HBM traffic of one column-kernel launch (FETCH_SIZE, WRITE_SIZE in separate --pmc passes, last dispatch of
# bench.py --steps 3): CFGS="<levels> ..."; LIB=<path> runs another build of the library (through
# MCKPP_HIP_LIBRARY: the product library is not touched); BENCH_ARGS e.g. "--ncol 7680" (one column per slot:
# no streaming traffic beside the passes).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/traffic
rm -rf $OUT && mkdir -p $OUT
if [ -n "$LIB" ]; then export MCKPP_HIP_LIBRARY=$(realpath $LIB); fi
for nz in ${CFGS:-60}; do
  B="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --nz $nz $BENCH_ARGS"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$nz -- $B > $OUT/fetch_$nz.json 2> $OUT/fetch_$nz.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$nz -- $B > $OUT/write_$nz.json 2> $OUT/write_$nz.err
  python3 - <<PY
import csv, glob
def last(tag):
    f = sorted(glob.glob("$OUT/%s_$nz/**/*counter_collection.csv" % tag, recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_column" in r["Kernel_Name"]]
    d = max(int(r["Dispatch_Id"]) for r in rows)
    return sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == d)
fe, wr = last("fetch"), last("write")
print("nz=$nz ${LIB:-product} FETCH_SIZE %.0f KB (x2: %.2f GB)  WRITE_SIZE %.0f KB (%.2f GB)  total %.2f GB" % (fe, 2 * fe * 1024 / 1e9, wr, wr * 1024 / 1e9, (2 * fe + wr) * 1024 / 1e9))
PY
done
