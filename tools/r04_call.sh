#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 600 $O/t_gru.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_gpu_principal_components.py -x -q -k "gru or token or artspeech or engine or full_size or raw or hidden"
tail -3 $O/t_gru.log
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 1 0 1 0 1 0; do
if [ $r = 1 ]; then export ARTSPEECH_GRU_SHARED_CUS=1; else unset ARTSPEECH_GRU_SHARED_CUS; fi
step 200 $O/bx_$r.log python bench.py $F
echo "shared $r: $(grep 'ms/step' $O/bx_$r.log)" | tee -a $O/gru_exclusive_ab.log
done
