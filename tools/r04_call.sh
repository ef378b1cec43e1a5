#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 600 $O/t_tr.log python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -k "engine or artspeech or train or full_size or run_epoch or heads"
tail -3 $O/t_tr.log
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 0 1 0 1 0 1; do
if [ $r = 1 ]; then export AS_NO_LOSS_TAIL=1; else unset AS_NO_LOSS_TAIL; fi
step 200 $O/bt_$r.log python bench.py $F
echo "separate loss kernel $r: $(grep 'ms/step' $O/bt_$r.log)" | tee -a $O/loss_tail_ab.log
done
unset AS_NO_LOSS_TAIL ARTSPEECH_DIAG_LIB
bash tools/timeline.sh loss
sed -n 16,22p $O/loss_timeline.txt
