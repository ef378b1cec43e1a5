#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/tri.log timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -k "triangular" || exit 1
tail -3 $O/tri.log
grep -q failed $O/tri.log && exit 1
step 900 $O/ttests.log python -m pytest tests/test_gpu_transformer.py tests/test_gpu_pipeline.py tests/test_gpu_train.py -q || exit 1
tail -2 $O/ttests.log
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
step 200 $O/att.log python tools/bench_attention.py
grep -v amdgpu $O/att.log | tail -6
