#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export ARTSPEECH_GEMM_PRECISION=lib
step 400 $O/fw_six.log python -m pytest tests/test_gpu_transformer.py -x -q -s -k "full_width_model_matches_reference_fixture"
grep -h "full-width transformer contours\|passed\|failed\|assert" $O/fw_six.log | head
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag_nine.so
step 400 $O/fw_nine.log python -m pytest tests/test_gpu_transformer.py -x -q -s -k "full_width_model_matches_reference_fixture"
grep -h "full-width transformer contours\|passed\|failed\|assert" $O/fw_nine.log | head
step 300 $O/bt_nine.log python tools/bench_transformer.py 32 200 4
grep "fwd+bwd\|forward" $O/bt_nine.log
