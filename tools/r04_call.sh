#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 400 $O/t_gru.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q -k "gru or artspeech or engine or full_size"
tail -2 $O/t_gru.log
step 120 $O/rm.log python3 tools/bench_gru.py 20
grep "gru fwd" $O/rm.log
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 0 1 0 1 0 1; do
if [ $r = 1 ]; then export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag_nogs.so; else unset ARTSPEECH_DIAG_LIB; fi
step 200 $O/bg_$r.log python bench.py $F
echo "no gate share $r: $(grep 'ms/step' $O/bg_$r.log)" | tee -a $O/gate_share_ab.log
done
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag_nogs.so
step 120 $O/rm0.log python3 tools/bench_gru.py 20
grep "gru fwd" $O/rm0.log
