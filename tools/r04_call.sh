#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag_nt.so
step 120 $O/hm_nt.log python3 tools/bench_heads.py 20
grep "head\." $O/hm_nt.log | head -14
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so
step 120 $O/hm_0.log python3 tools/bench_heads.py 20
grep "head\." $O/hm_0.log | head -14
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 0 1 0 1 0 1; do
if [ $r = 1 ]; then export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag_nt.so; else export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so; fi
step 200 $O/bn_$r.log python bench.py $F
echo "non-temporal $r: $(grep 'ms/step' $O/bn_$r.log)"
done
