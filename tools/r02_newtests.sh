cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "zero_pivot or long_iteration or tolerance or restart_and_ancillary" 2>&1 | grep -v amdgpu.ids | grep -B5 -A25 "Error\|assert" | head -80
