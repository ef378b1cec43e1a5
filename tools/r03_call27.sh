#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 600 $O/att_tests.log python -m pytest tests/test_gpu_transformer.py -x -q || exit 1
tail -2 $O/att_tests.log
grep -q failed $O/att_tests.log && exit 1
step 300 $O/bench_attention.log python tools/bench_attention.py
tail -12 $O/bench_attention.log
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
