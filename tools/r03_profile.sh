# Round-3 profiles (run on the GPU box through gpurun): kernel trace + stats of the default bench run,
# then SQ and TCC counter passes (separate --pmc runs, kernel-trace only) for three shapes.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "stats done"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
for nz in 60 69 100; do
  B="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --nz $nz"
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/sq1_$nz -- $B > $OUT/sq1_$nz.json 2> $OUT/sq1_$nz.err
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/sq2_$nz -- $B > $OUT/sq2_$nz.json 2> $OUT/sq2_$nz.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$nz -- $B > $OUT/fetch_$nz.json 2> $OUT/fetch_$nz.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$nz -- $B > $OUT/write_$nz.json 2> $OUT/write_$nz.err
  echo "pmc nz=$nz done"
done
python3 tools/r03_profile_digest.py $OUT
# keep only the small files for the merge back
find $OUT -name "*.csv" -size +3M -delete
du -sh $OUT
