#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/hidden_sizes.log timeout -k 10 280 python tools/bench_hidden_sizes.py
cat $O/hidden_sizes.log | grep -v amdgpu.ids
step 900 $O/tests.log python -m pytest tests -m gpu -q || exit 1
tail -2 $O/tests.log
