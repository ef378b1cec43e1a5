#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 600 $O/parity.log python -m pytest tests/test_gpu_parity.py -x -q
tail -3 $O/parity.log
ARTSPEECH_MATRIX_ARITH=fp32 step 300 $O/wgrad_fp32.log python tools/bench_wgrad.py
step 300 $O/wgrad_s6.log python tools/bench_wgrad.py
echo "--- fp32"; cat $O/wgrad_fp32.log | tail -15; echo "--- s6"; cat $O/wgrad_s6.log | tail -15
