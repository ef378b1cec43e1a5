#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 1100 $O/t_all.log python -m pytest tests -x -q -m gpu
tail -5 $O/t_all.log
