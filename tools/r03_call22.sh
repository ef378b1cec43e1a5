#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 200 $O/gemm_ext_bench.log python tools/bench_gemm_ext.py 10
cat $O/gemm_ext_bench.log
