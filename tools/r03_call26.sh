#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 600 $O/sk_tests.log python -m pytest tests/test_gpu_parity.py -x -q -k "stream_k or split_k or wgrad or gemm" || exit 1
tail -3 $O/sk_tests.log
grep -q passed $O/sk_tests.log || exit 1
grep -q failed $O/sk_tests.log && exit 1
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
step 900 $O/tests.log python -m pytest tests -m gpu -q || exit 1
tail -3 $O/tests.log
