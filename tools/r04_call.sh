#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 0 1 0 1 0 1 0 1; do
if [ $r = 1 ]; then export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag_nogs.so; else export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so; fi
step 200 $O/bg_$r.log python bench.py $F
echo "no gate share $r (both arms: diagnostic flavour): $(grep 'ms/step' $O/bg_$r.log)" | tee -a $O/gate_share_ab.log
done
