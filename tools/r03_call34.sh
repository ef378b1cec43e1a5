#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/gru_tests.log timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -k "gru_layer or other_hidden or bad_arguments" || exit 1
tail -3 $O/gru_tests.log
grep -q failed $O/gru_tests.log && exit 1
step 300 $O/hidden_sizes.log timeout -k 10 280 python tools/bench_hidden_sizes.py
cat $O/hidden_sizes.log | grep -v amdgpu.ids
