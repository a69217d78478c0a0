cd $GRAFT_REPO_ROOT
( time timeout -k 10 900 python -m pytest tests/test_fortran_host.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | tail -30
