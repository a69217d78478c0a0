# Round-5 profiles (on the GPU box through gpurun): kernel trace + stats of the default bench run (headline
# kernel: k_column_ps<0,0>), then SQ and TCC counter passes (separate --pmc runs, kernel-trace only) for the
# headline shape in both solver modes, the shipped-namelist shape (69 stretched levels, 35 % land, dto 1200) and
# 100 levels.  Digest -> gpurun_out/prof_r05/counters.json.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r05
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "stats done"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
for cfg in 60:0 60:1 69s:0 100:0 100:1; do
  IFS=: read shape sm <<< "$cfg"
  unset MCKPP_PS_CONFLICT_FREE
  if [ "$shape" = "69s" ]; then args="--nz 69 --grid stretched --dto 1200 --land 0.35";
  elif [ "$shape" = "69sfree" ]; then args="--nz 69 --grid stretched --dto 1200 --land 0.35"; export MCKPP_PS_CONFLICT_FREE=1;   # the slot stride without bank conflicts (experiment)
  else args="--nz $shape"; fi
  export MCKPP_SOLVER_MODE=$sm
  B="python3 bench.py --steps 3 --warmup 2 --settle 0 --no-cpu-baseline --no-extras $args"
  t=${shape}_sm$sm
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/sq1_$t -- $B > $OUT/sq1_$t.json 2> $OUT/sq1_$t.err
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/sq2_$t -- $B > $OUT/sq2_$t.json 2> $OUT/sq2_$t.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$t -- $B > $OUT/fetch_$t.json 2> $OUT/fetch_$t.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$t -- $B > $OUT/write_$t.json 2> $OUT/write_$t.err
  echo "pmc $t done"
done
unset MCKPP_SOLVER_MODE MCKPP_PS_CONFLICT_FREE
python3 tools/r05_profile_digest.py $OUT
find $OUT -name "*.csv" -size +3M -delete
du -sh $OUT
