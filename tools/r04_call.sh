#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
export ARTSPEECH_DIAG_LIB=1
run() { step 200 $O/v.log python bench.py --no-extras --no-cpu-baseline --no-profile --no-exact; grep '^{' $O/v.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for rep in 1 2; do
run "all-s6"
AS_NO_LIN_OUT_S6=1 run "out-fp32"
AS_NO_PLAIN_S6=1 run "dx1-fp32"
AS_NO_LIN_OUT_S6=1 AS_NO_PLAIN_S6=1 run "out+dx1-fp32"
ARTSPEECH_MATRIX_ARITH=fp32 run "all-fp32"
done
