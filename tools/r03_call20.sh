#!/bin/bash
# round 3, call 20: extended GEMM epilogue / segmented reduction + the fused ChannelBlocks node of the transformer
source tools/gpu_steps.sh
O=gpurun_out/r03t
mkdir -p $O
step 300 $O/gemm_ext.log python -m pytest tests/test_gpu_parity.py -x -q -k "gemm" || exit 1
step 900 $O/transformer_tests.log python -m pytest tests/test_gpu_transformer.py -x -q || exit 1
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
