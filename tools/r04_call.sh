#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 0 1 0 1 0 1 0 1; do
if [ $r = 1 ]; then export AS_CHAIN_JOIN=1; else unset AS_CHAIN_JOIN; fi
step 200 $O/bj_$r.log python bench.py $F
echo "chain join $r: $(grep 'ms/step' $O/bj_$r.log)" | tee -a $O/join_ab.log
done
unset AS_CHAIN_JOIN ARTSPEECH_DIAG_LIB
bash tools/timeline.sh joins
tail -6 $O/joins_timeline.txt
