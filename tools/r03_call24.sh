#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 200 $O/gemm_ext_bench.log python tools/bench_gemm_ext.py 10
cat $O/gemm_ext_bench.log
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
step 900 $O/tests.log python -m pytest tests -m gpu -q || exit 1
tail -3 $O/tests.log
