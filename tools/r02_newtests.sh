cd $GRAFT_REPO_ROOT
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "seeded or edge_sizes" ) 2>&1 | grep -v amdgpu.ids | tail -30
