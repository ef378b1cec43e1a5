#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/gemm_ext.log python -m pytest tests/test_gpu_parity.py -x -q -k "gemm" || exit 1
tail -3 $O/gemm_ext.log
step 200 $O/gemm_ext_bench.log python tools/bench_gemm_ext.py 10
cat $O/gemm_ext_bench.log
step 900 $O/transformer_tests.log python -m pytest tests/test_gpu_transformer.py -x -q || exit 1
tail -3 $O/transformer_tests.log
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
