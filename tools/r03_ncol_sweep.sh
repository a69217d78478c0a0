cd $GRAFT_REPO_ROOT
for n in 92160 99840 100000 101000 103680 107520 115200; do
  python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --nz 60 --ncol $n 2>/dev/null | sed -e "s/.*\"value\": \([0-9.e+]*\).*\"ms_per_step\": \([0-9.]*\).*/ncol=$n rate \1 column-steps\/s, \2 ms per step/"
done
