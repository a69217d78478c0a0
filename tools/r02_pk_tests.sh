set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02pk
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "pk" > gpurun_out/r02pk/tests.log 2>&1 || { tail -40 gpurun_out/r02pk/tests.log; exit 1; }
tail -3 gpurun_out/r02pk/tests.log
for g in 4x4 5x3 6x2 7x2 8x2; do
MCKPP_PK=$g timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "pk" > gpurun_out/r02pk/tests_$g.log 2>&1 || { tail -40 gpurun_out/r02pk/tests_$g.log; exit 1; }
echo $g; tail -1 gpurun_out/r02pk/tests_$g.log
done
