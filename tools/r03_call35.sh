#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/ln.log timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_row" || exit 1
tail -3 $O/ln.log
grep -q failed $O/ln.log && exit 1
step 900 $O/tests.log python -m pytest tests -m gpu -q || exit 1
tail -2 $O/tests.log
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
