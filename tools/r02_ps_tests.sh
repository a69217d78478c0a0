cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "ps" 2>&1 | grep -v amdgpu.ids | tail -25
