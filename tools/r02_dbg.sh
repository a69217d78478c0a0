cd $GRAFT_REPO_ROOT
MCKPP_KERNEL=pk timeout -k 10 120 python tools/dbg_trap.py 40 2>&1 | grep -v amdgpu.ids
MCKPP_KERNEL=pk MCKPP_PK=4x4 timeout -k 10 120 python tools/dbg_trap.py 40 2>&1 | grep -v amdgpu.ids
