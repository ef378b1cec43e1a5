#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/gru_tests.log timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -k "gru_layer" || exit 1
tail -3 $O/gru_tests.log
grep -q failed $O/gru_tests.log && exit 1
step 300 $O/gru_model_tests.log timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -k "other_hidden" || exit 1
tail -15 $O/gru_model_tests.log
