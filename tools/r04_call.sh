#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 400 $O/t_wg.log python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient_split or residual_mask or split_matrix"
tail -3 $O/t_wg.log
step 700 $O/t_tr.log python -m pytest tests/test_gpu_transformer.py -x -q
tail -3 $O/t_tr.log
step 200 $O/lin.log python tools/bench_linear.py 20
grep -A3 "x 110\|big:\|Linear x 8" $O/lin.log | grep "in-kernel\|M="
step 300 $O/bt_lib.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd" $O/bt_lib.log
export ARTSPEECH_GEMM_PRECISION=lib
step 300 $O/bt_all.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd" $O/bt_all.log
