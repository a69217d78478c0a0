cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "verticalmixing or fortran" 2>&1 | grep -v amdgpu.ids | tail -30
