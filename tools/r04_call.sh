#!/bin/bash
# scratch: one gpurun call of round 4
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out
step 400 $O/t_pc.log python -m pytest tests/test_gpu_principal_components.py -x -q
tail -5 $O/t_pc.log
step 300 $O/bt_lib.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd" $O/bt_lib.log
export ARTSPEECH_GRAD_PRECISION=f32
step 300 $O/bt_f32.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd" $O/bt_f32.log
export ARTSPEECH_GRAD_PRECISION=lib
export ARTSPEECH_GEMM_PRECISION=lib
step 300 $O/bt_all.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd" $O/bt_all.log
