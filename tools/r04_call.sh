#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 600 $O/t_tr.log python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -k "engine or artspeech or train or full_size or run_epoch"
tail -3 $O/t_tr.log
bash tools/timeline.sh loss
sed -n 16,24p $O/loss_timeline.txt
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
step 200 $O/bl.log python bench.py $F
grep "ms/step" $O/bl.log
