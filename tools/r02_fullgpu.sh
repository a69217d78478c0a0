cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02full
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02full/tests.log 2>&1
rc=$?
tail -30 gpurun_out/r02full/tests.log
exit $rc
