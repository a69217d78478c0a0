# k_column_ps: its own parity cases, then the optional-physics suite and the rest of the GPU suite with
# MCKPP_KERNEL=ps forced (tests that pin a kernel themselves keep their choice).
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "ps" 2>&1 | grep -v amdgpu.ids | tail -8 &&
MCKPP_KERNEL=ps timeout -k 10 900 python -m pytest tests/test_options_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8
