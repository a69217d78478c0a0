cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02full
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02full/tests.log 2>&1
rc=$?
tail -6 gpurun_out/r02full/tests.log
exit $rc
