#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export AS_FUZZ_SEEDS=120
step 1100 $O/t_fuzz.log python -m pytest tests -x -q -m gpu -k "random_configurations or random_shapes or sweep"
tail -8 $O/t_fuzz.log
