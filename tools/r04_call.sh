#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 200 $O/lin.log python tools/bench_linear.py 20
grep -A1 "x 110\|big:\|Linear x 8" $O/lin.log
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so AS_NO_XCD_PANELS=1
step 200 $O/lin0.log python tools/bench_linear.py 20
grep -A1 "x 110\|big:\|Linear x 8" $O/lin0.log
unset AS_NO_XCD_PANELS ARTSPEECH_DIAG_LIB
step 300 $O/bt_lib.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd\|forward" $O/bt_lib.log
step 600 $O/t_tr.log python -m pytest tests/test_gpu_transformer.py -x -q
tail -3 $O/t_tr.log
