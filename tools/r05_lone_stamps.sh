# Phase stamps of a lone column's pass (tools/r05_lone_probe.py under a -DMCKPP_PS_STAMPS build, .ab/libS.so)
cd $GRAFT_REPO_ROOT
export MCKPP_HIP_LIBRARY=$PWD/.ab/libS.so MCKPP_STAMP=1 MCKPP_PS_VERBOSE=1
timeout -k 10 500 python tools/r05_lone_probe.py ${NZS:-60 100} 2>&1
