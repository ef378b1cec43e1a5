#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
F="--no-extras --no-cpu-baseline --no-profile --no-exact"
for r in 0 64 32 64 0; do
export ARTSPEECH_SIDE_RESERVE_CUS=$r
step 200 $O/bm_$r.log python bench.py $F
echo "reserve $r: $(grep 'ms/step' $O/bm_$r.log)"
done
export ARTSPEECH_SIDE_RESERVE_CUS=64
step 120 $O/rs64.log python3 tools/recurrence_stamps.py 50
grep -v amdgpu $O/rs64.log | tail -4
bash tools/timeline.sh mask64 ARTSPEECH_SIDE_RESERVE_CUS=64
cat $O/mask64_timeline.txt
