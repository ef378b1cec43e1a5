#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03f; mkdir -p $out
step 300 $out/tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_train.py -m gpu -q -x
tail -5 $out/tests.log
step 120 $out/metrics_kernels.log python tools/bench_metrics_kernels.py 50 --json $out/metrics_kernels.json
grep -v amdgpu $out/metrics_kernels.log
