#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 1000 $O/gpu_tests.log python -m pytest tests -x -q -m gpu
tail -3 $O/gpu_tests.log
step 400 $O/tbench.log python tools/bench_transformer.py 32 200 3
grep -v amdgpu.ids $O/tbench.log | tail -4
ARTSPEECH_MATRIX_ARITH=fp32 step 400 $O/tbench_fp32.log python tools/bench_transformer.py 32 200 3
grep -v amdgpu.ids $O/tbench_fp32.log | tail -3
