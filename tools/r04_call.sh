#!/bin/bash
# one gpurun call of round 4 (rewritten per call; see tools/gpu_steps.sh)
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 600 $O/parity_heads.log python -m pytest tests/test_gpu_parity.py -x -q -k "head or predictor or fixture or artspeech"
ARTSPEECH_MATRIX_ARITH=fp32 step 200 $O/heads_fp32.log python tools/bench_heads.py 20
step 200 $O/heads_s6.log python tools/bench_heads.py 20
tail -3 $O/parity_heads.log; cat $O/heads_fp32.log $O/heads_s6.log
