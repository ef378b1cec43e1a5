#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 300 $O/graph.log python tools/bench_graph.py 30
grep -v amdgpu $O/graph.log | tail -6
step 400 $O/t_tr.log python -m pytest tests/test_gpu_train.py -x -q
tail -2 $O/t_tr.log
